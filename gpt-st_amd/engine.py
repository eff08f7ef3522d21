"""Hand-written forward / backward of the GPT-ST pretraining network on the HIP kernels (no torch autograd inside).

Every ``*_fwd`` returns ``(outputs..., saved)`` and every ``*_bwd`` ACCUMULATES parameter gradients into the tensors of a
gradient dict ``g`` (views into one flat buffer) and returns the input gradients.  ``p`` / ``g`` map the reference's
state_dict keys to tensors.  Citations: reference model/Pretrain_model/GPTST.py.
"""
import contextlib
import os
import threading

import torch

from . import _C, ops
from .ops import (EPI_ADD_DPRE, EPI_ADD_PREMUL, EPI_LRELU, EPI_PREMUL, EPI_PLAIN, EPI_RES_LRELU, MODE_NODE, MODE_SHARED, MODE_TIME, PRO_DPRE, PRO_NONE)

TF_NAMES = ("ln_day", "ln_week", "ln1", "ln2", "ln")


class ZeroArena:
    """All zero-initialised scratch of one step (bias-gradient accumulators, embedding gradients) comes from ONE buffer that
    is cleared with ONE memset per step instead of ~30 torch.zeros launches.  The first (sizing) pass falls back to torch.zeros."""

    def __init__(self, device):
        self.device, self.buf, self.off, self.need = device, None, 0, 0

    def begin(self, zero=True):
        """-> the buffer the caller has to zero itself when zero=False (None when there is none or it was just created zeroed)"""
        pending = None
        if self.buf is None or self.buf.numel() < self.need:
            self.buf = torch.zeros(max(self.need, 1), device=self.device) if self.need else None
        elif self.buf is not None:
            if zero:
                self.buf.zero_()
            else:
                pending = self.buf
        self.off, self.need = 0, 0
        return pending

    def zeros(self, *shape):
        n = 1
        for d in shape:
            n *= d
        n4 = (n + 3) // 4 * 4
        self.need += n4
        if self.buf is None or self.off + n4 > self.buf.numel():
            return torch.zeros(*shape, device=self.device)
        t = self.buf[self.off:self.off + n].view(*shape)
        self.off += n4
        return t


class _Context(threading.local):
    """Per-thread execution context of a step (thread-local so that several shard steppers can run side by side in one process:
    tests emulate the ranks of a node-sharded run with threads).
      ARENA        ZeroArena of the running step (None -> plain torch.zeros)
      NODE_REDUCE  node-sharded run: callable that completes a sum over nodes across the ranks, in place (None -> single shard)
      NO_HANDOFFS  the step is enqueued without launches in which one workgroup waits for another (no_handoffs(): a stepper in safe mode) — per
                   thread, so that one emulated rank in safe mode does not change the launches another thread is enqueuing"""
    ARENA = None
    NODE_REDUCE = None
    NO_HANDOFFS = False


CTX = _Context()


# (r05: the side-stream modes of rounds 1-3 — weight gradients, reductions, parameter generation and the KL backward on a second stream / graph
#  branch — were each measured slower than the single chain (DESIGN.md sections 7-8: 364 vs 403, 558-579 vs 597, 651 / 659 vs 675 steps/s) and left
#  the engine; the one fork that remains is the data-parallel gradient bucket's all-reduce, Reductions.flush_async.)


class Reductions:
    """Everything of a backward pass that only produces PARAMETER gradients from per-layer partial results — the sums of the
    generated weights' gradients into their pools and embeddings (GPTST.py:24-25,29-30,104,129,137-138,156,160-161 backward), the
    temporal-graph factor gradients and the seven time-feature MLPs — is collected here while the data-gradient chain runs and
    executed at the end of the backward as THREE launches (gram_bwd, one job table, one time-feature job table) instead of ~20
    small launches per STHCN.  Nothing on the critical chain waits for these results; only the optimiser does."""

    def __init__(self):
        self.jobs = ops.PoolJobs()   # reductions whose inputs are complete when they are queued (their producer has been launched)
        self.late = ops.PoolJobs()   # reductions over dA, which gram_bwd writes in _run(): never carried by an earlier launch
        self.no_carry = False        # (deterministic steppers: one reduction launch, as the single-owner kind-2 launch groups by target)
        self.grams = []          # (A (L*N,Hm,T), dG (L*N,T,T), dA out)
        self.tf = []             # (params, grads, dout, rows, K)
        self.keep = []
        self.on_bucket = None    # callable(k): gradient bucket k is complete (k = 0: the decoder's) — called on the forked stream, right behind
        self.nbucket = 0         # the reductions that finish it, so a data-parallel step can enqueue that bucket's all-reduce under the rest of the backward

    def take_carry(self, mb):
        """Up to `mb` MB of the queued reductions for a backward launch that carries them as role workgroups (ops.cap_cross_route_lin_bwd jobs=): the
        decoder's reductions under the encoder's routing backward.  None: nothing to carry / not in this step (data-parallel bucket overlap: the bucket's
        reductions run where the bucket closes; node shards; deterministic mode)."""
        if not CARRY_RED or self.no_carry or self.on_bucket is not None or CTX.NODE_REDUCE is not None or not self.jobs.jobs:
            return None
        take, rest, nb = [], [], 0.0
        for j in self.jobs.jobs:
            b = 4.0 * j[5] * j[7] * j[8] / 2 ** 20                # R * cols * nsplit floats
            if j[0] in (ops.PoolJobs.BWD_POOL, ops.PoolJobs.BWD_EMB) and nb + b <= mb and len(take) < 100 and (j[7] | j[9]) % 4 == 0:     # (float4-shaped jobs only)
                take.append(j); nb += b
            else:
                rest.append(j)
        if not take:
            return None
        self.jobs.jobs = rest
        c = ops.PoolJobs()
        c.jobs = take
        return c

    def untake(self, carry):
        if carry is not None and carry.jobs:
            self.jobs.jobs = carry.jobs + self.jobs.jobs
            carry.jobs = []

    def gram(self, A, dG, dA, N, nsG):
        """A (L*N,Hm,T), dG (L, nsG, N, T, T) partial graph gradients, dA (L*N,Hm,T) output"""
        self.grams.append((A, dG, dA, N, nsG))

    def timefeat(self, p, g, pfx, tidx, dout, spg=False):
        B, T = tidx.shape[0], tidx.shape[1]
        rows, K = (B, T) if spg else (B * T, 1)
        self.tf.append((_tf_tensors(p, pfx), _tf_tensors(g, pfx), dout, rows, K))

    def flush_async(self, tidx):
        """A gradient bucket is complete (a data-parallel step, GPTST_DP_OVERLAP=1; no-op otherwise): its reductions run HERE, on the calling
        stream (as a side branch their ~1300 bandwidth-bound workgroups slowed the chain they ran under by more than they hid: 717 vs 752
        steps/s at one rank), and only the bucket's all-reduce — a few RCCL workgroups — is forked under the rest of the backward."""
        if getattr(self, "bucket_inline", False) and self.on_bucket is not None and self.nbucket == 0 and (self.jobs.jobs or self.late.jobs or self.grams or self.tf):
            self._run(tidx)
            self.fork_side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.fork_side):
                self.on_bucket(0)
            self.forked = True
            self.nbucket += 1

    def flush(self, tidx):
        self._run(tidx)
        if getattr(self, "forked", False):
            torch.cuda.current_stream().wait_stream(self.fork_side)
            self.forked = False

    def _run(self, tidx):
        gr, self.grams = sorted(self.grams, key=lambda t: t[0].data_ptr()), []
        while gr:                                   # adjacent (A, dG, dA) triples (both STHCNs of a step) share one launch
            A, dG, dA, N, nsG = gr.pop(0)
            while gr and gr[0][0].data_ptr() == A.data_ptr() + 4 * A.numel() and gr[0][1].data_ptr() == dG.data_ptr() + 4 * dG.numel() \
                    and gr[0][2].data_ptr() == dA.data_ptr() + 4 * dA.numel() and gr[0][3:] == (N, nsG):
                A2, dG2, dA2 = gr.pop(0)[:3]
                A = torch.as_strided(A, (A.shape[0] + A2.shape[0],) + tuple(A.shape[1:]), A.stride())
                dG = torch.as_strided(dG, (dG.shape[0] + dG2.shape[0],) + tuple(dG.shape[1:]), dG.stride())
                dA = torch.as_strided(dA, (dA.shape[0] + dA2.shape[0],) + tuple(dA.shape[1:]), dA.stride())
            ops.gram_bwd(A, dG, out=dA, layers=A.shape[0] // N, nsplit=nsG)
        self.jobs.jobs, self.late.jobs = self.jobs.jobs + self.late.jobs, []
        self.jobs.launch()
        tf, self.tf = self.tf, []
        ops.timefeat_jobs_bwd(tf, tidx)
        self.keep = []


def _zeros(ref, *shape):
    if CTX.ARENA is not None:
        return CTX.ARENA.zeros(*shape)
    return torch.zeros(*shape, device=ref.device)


def _tf_tensors(d, pfx):
    out = []
    for n in TF_NAMES:
        out += [d[pfx + n + ".weight"], d[pfx + n + ".bias"]]
    return out


# ---- time features (GPTST.py:187-219) -----------------------------------------------------------------------------
def _tf_job(p, pfx, tidx, spg=False):
    B, T = tidx.shape[0], tidx.shape[1]
    rows, K = (B, T) if spg else (B * T, 1)
    return (_tf_tensors(p, pfx), rows, K)


# ---- hyperTem (GPTST.py:154-163) -----------------------------------------------------------------------------------
# Parameter generation (A, G, W_bt, b_bt) and the gradient reductions into the pools / embeddings are batched per STHCN
# (sthcn_fwd / sthcn_bwd below): one launch for all four hyperTem layers instead of one per layer.
# C = 64: the forward may skip writing R = G (*) X and the fused backward rebuild it from X (12 L2-hit float4 per operand).  Measured r03 at the
# bench shape (scratch/mb_ht5.py): forward 22.2 -> 18.8 us, backward 29.4 -> 37.9 us (the weight-gradient workgroups become the long pole: 565 KB
# of L2 reads each) — a net loss of 5 us per layer although it removes 33 MB of HBM traffic, so R is kept.  GPTST_DROP_R=1 switches it on.
DROP_R = os.environ.get("GPTST_DROP_R", "0") == "1"


def _ht_fused_bwd(dims):
    """the one-launch hyperTem backward (hypertem_bwd_wgrad) serves this step"""
    return dims[3] == 64 and FUSE_HT_BWD


def hypertem_core_fwd(x, G, Wbt, bbt, dims):
    """x (BTN, C) rows -> out (BTN, C);  G (N,T,T), Wbt (BT,C,C), bbt (BT,C) precomputed."""
    B, T, N, C = dims
    if C == 64:
        keep = not (DROP_R and _ht_fused_bwd(dims) and ops.wgrad_nsplit(MODE_TIME, B * T, N, C) == 1)
        R, out = ops.hypertem_fwd(x.view(B, T, N, C), G, Wbt, bbt, want_R=keep)         # :157-158 + :162-163 fused
        R, out = (R.view(-1, C) if keep else None), out.view(-1, C)
    else:
        R = ops.tmix(x.view(B, T, N, C), G).view(-1, C)                                 # :157-158
        out = ops.apply(R, Wbt, MODE_TIME, B * T, N, bias=bbt, resid=x, epi=EPI_RES_LRELU)  # :162-163
    return out, (x, R, out, G, Wbt)


FUSE_HT_BWD = True         # hyperTem backward + its weight gradient in one launch (False: two launches)


def graph_grad_splits(dims):
    """partial graph gradients per layer: the fused C = 64 backward writes one per sample, the unfused path one in total"""
    return dims[0] if dims[3] == 64 else 1


CHAIN128 = os.environ.get("GPTST_CHAIN128", "1") == "1"


def chain_ok(dims):
    """The "dPre chain" (include/gptst_hip.h, gptst_hypertem_bwd): every backward kernel of the layer chain hands its input gradient down
    already multiplied by lrelu'(its input), so that no kernel reads its own output only for the sign.  C = 64: the fused kernels; C = 128
    (r05): the unfused passes in their chain forms (apply128 / wgrad128 without the dPre prologue, gptst_tmix_bwd_chain, epilogue 4) — three
    activation-sized reads less per hyperTem layer, two per cap (BASELINE configs[4])."""
    return _ht_fused_bwd(dims) or (CHAIN128 and dims[3] == 128)


def hypertem_core_bwd(saved, dout, dG_out, dims, chain=False, premul=False):
    """-> dx, (dWbt, nsplit, (dbias partials, their count)); the graph-gradient partials are written into dG_out (nsG, N, T, T).
    chain: dout already is dPre (the layer's output is not read);  premul (chain only): dx is returned multiplied by lrelu'(x)."""
    B, T, N, C = dims
    x, R, out, G, Wbt = saved
    BT = B * T
    assert not chain or chain_ok(dims)
    if _ht_fused_bwd(dims):
        # data / graph gradients and the weight + bias gradient side by side in one launch: rows [dW_bt | db_bt]
        dx, dWb, ns, _ = ops.hypertem_bwd_wgrad(dout.view(B, T, N, C), None if chain else out.view(B, T, N, C), x.view(B, T, N, C), G, Wbt,
                                                R.view(B, T, N, C) if R is not None else None, dG=dG_out, premul=chain and premul)
        dWbt, dbias, nsb = dWb[:, :C * C], dWb[:, C * C:], ns
        dx = dx.view(-1, C)
    elif C == 64:
        # the weight-gradient kernel also emits the bias gradient (column sums of dPre per (b,t)): rows [dW_bt | db_bt]
        dWb, ns = ops.wgrad(R, dout, MODE_TIME, BT, N, D2=out, pro=PRO_DPRE, colsum_d=True)
        dWbt, dbias, nsb = dWb[:, :C * C], dWb[:, C * C:], ns
        dx, _, _ = ops.hypertem_bwd(dout.view(B, T, N, C), out.view(B, T, N, C), x.view(B, T, N, C), G, Wbt, dG=dG_out, want_dbias=False)
        dx = dx.view(-1, C)
    elif chain:                                       # C = 128, dPre chain: no pass reads the layer's output
        dWb, ns = ops.wgrad(R, dout, MODE_TIME, BT, N, colsum_d=True)
        dWbt, dbias, nsb = dWb[:, :C * C], dWb[:, C * C:], ns
        dR = ops.apply(dout, Wbt, MODE_TIME, BT, N, transw=True)
        dx, _ = ops.tmix_bwd_chain(dR.view(B, T, N, C), x.view(B, T, N, C), G, dout.view(B, T, N, C), premul=premul, dG=dG_out[0])
        dx = dx.view(-1, C)
    else:
        dWb, ns = ops.wgrad(R, dout, MODE_TIME, BT, N, D2=out, pro=PRO_DPRE, colsum_d=True)
        dWbt, dbias, nsb = dWb[:, :C * C], dWb[:, C * C:], ns
        dR = ops.apply(dout, Wbt, MODE_TIME, BT, N, A2=out, transw=True, pro=PRO_DPRE)
        dx, _ = ops.tmix_bwd(dR.view(B, T, N, C), x.view(B, T, N, C), G, dout.view(B, T, N, C), out.view(B, T, N, C), dG=dG_out[0])
        dx = dx.view(-1, C)
    return dx, (dWbt, ns, (dbias, nsb))


CARRY_RED = os.environ.get("GPTST_CARRY_RED", "1") == "1"        # queued weight-gradient reductions as role workgroups of the routing backward (r05, late)
CARRY_MB = float(os.environ.get("GPTST_CARRY_MB", "25"))        # at most this much of them per launch
PAIR_UNDER_DP = os.environ.get("GPTST_PAIR_UNDER_DP", "1") == "1"    # the decoder-hyperTem1 / encoder-hyperTem4 pair also when the decoder's gradient bucket leaves early (r05)
PAIR_BWD = os.environ.get("GPTST_PAIR_BWD", "1") == "1"        # two adjacent hyperTem layers' backward in one launch (r04)


@contextlib.contextmanager
def no_handoffs():
    """Enqueue (or capture) the step without launches in which one workgroup waits for another (the hyperTem backward pairs, the cross-time role
    of the routing backward, the cooperative mask launch): what a stepper falls back to after a bounded wait expired (step.py::_enter_safe_mode).
    Everything it switches is per thread (CTX, the library's thread-local mask switch) and put back as it was found."""
    from . import _C
    keep = (CTX.NO_HANDOFFS, _C.lib().value("gptst_mask_cooperative_state"))
    CTX.NO_HANDOFFS = True
    _C.lib().call("gptst_mask_cooperative", 0)
    try:
        yield
    finally:
        CTX.NO_HANDOFFS = keep[0]
        _C.lib().call("gptst_mask_cooperative", keep[1])


def pair_bwd_on():
    return PAIR_BWD and not CTX.NO_HANDOFFS


def cross_role():
    return 0 if CTX.NO_HANDOFFS else CROSS_ROLE


def _ht_pair_shape_ok(dims):
    """the launcher's own shape conditions (gptst_hypertem_bwd_pair, hypertem.hip): T = 12, and rounding a split's rows up to even must not
    change the split count — checked HERE so that a deferred hyperTem1 (PendingH1) is never handed to a launch that would refuse it"""
    B, T, N, C = dims
    if T != 12 or C != 64:
        return False
    ns = ops.wgrad_nsplit(MODE_TIME, B * T, N, C)
    rps = (-(-N // ns) + 1) & ~1
    return -(-N // rps) == ns


def ht_pair_ok(saved1, saved0, dims):
    return (pair_bwd_on() and dims[3] == 64 and not isinstance(saved1, EncIn) and saved1[1] is not None and saved0[1] is not None
            and _ht_fused_bwd(dims) and _ht_pair_shape_ok(dims))


class PendingH1:
    """the decoder's hyperTem1 backward, deferred into the pair launch with the encoder's hyperTem4 (GPTST.py:271 -> :454): its dPre, saved
    tensors and the output buffers the decoder's reduction jobs already point at"""

    def __init__(self, saved, dd, dG, dWb):
        self.saved, self.dd, self.dG, self.dWb = saved, dd, dG, dWb


def ht_pair_bwd(saved1, saved0, dout, dG1, dG0, dims, dWb1=None):
    """dPre-chain backward of hyperTem layers 1 (upper, dout = its dPre) and 0 in ONE launch -> dx0 (times lrelu'(x0)), hp1, hp0 (as
    hypertem_core_bwd's second result), or None (shape / no saved R)."""
    B, T, N, C = dims
    if not ht_pair_ok(saved1, saved0, dims):
        assert dWb1 is None
        return None
    x1, R1, _, G1, Wbt1 = saved1
    x0, R0, _, G0, Wbt0 = saved0
    v = lambda a: a.view(B, T, N, C)
    r = ops.hypertem_bwd_pair(v(dout), v(x1), G1, Wbt1, v(R1), v(x0), G0, Wbt0, v(R0), dG1, dG0, _zeros(dout, B), dWb1=dWb1)
    assert r is not None or dWb1 is None
    if r is None:
        return None
    dmid, dx0, dWb1, dWb0, ns = r
    return dx0.view(-1, C), (dWb1[:, :C * C], ns, (dWb1[:, C * C:], ns)), (dWb0[:, :C * C], ns, (dWb0[:, C * C:], ns))


# ---- cap (GPTST.py:100-141) ----------------------------------------------------------------------------------------
FUSE_CROSS = os.environ.get("GPTST_FUSE_CROSS", "1") == "1"     # cross-time block folded into its (b,t)-grouped neighbours (r03)
CAP_LIN = os.environ.get("GPTST_CAP_LIN", "1") == "1"           # ... and the entry Linear's backward folded into the same launch (r05)
CROSS_ROLE = int(os.environ.get("GPTST_CROSS_ROLE", "1"))       # ... its backward as a ROLE of the routing backward's launch (r04; 0: replicated prologue —
                                                                # what a stepper falls back to after a lost hand-off.  The rec backward as a THIRD role
                                                                # measured 799 vs 814 steps/s, profiles/r04_roles3_stamps.txt, and left the library in r05)


def cap_head_fwd(p, pfx, x, dadj, dyn, dims, num_route, HS, HT):
    """cap up to the cluster -> node scatter (GPTST.py:102-135): x (BTN,C) -> rec (BTN,C), c (BT,HS,N), (s, v, Ht, Rt, Y)."""
    B, T, N, C = dims
    c, s, Y = ops.cap_route_fwd(x.view(B, T, N, C), p[pfx + "ln_p.weight"], p[pfx + "ln_p.bias"], dadj, HS, num_route,
                                reduce_nodes=CTX.NODE_REDUCE, want_Y=True)                                            # :102-123
    fused = ops.cap_cross_rec_fwd(s, dyn, p[pfx + "mask_template"], c, B, T, N, HS, HT) if (FUSE_CROSS and CTX.NODE_REDUCE is None) else None
    if fused is not None:                                                                                             # :125-135, one launch
        v, Ht, Rt, rec = fused
    else:
        v, Ht, Rt = ops.cap_cross_fwd(s, dyn, p[pfx + "mask_template"], B, T, HS, HT)                                 # :125-134
        rec = ops.cap_rec_fwd(c, v, N, C)                                                                             # :135
    return rec, c, (s, v, Ht, Rt, Y)


def cap_core_fwd(p, pfx, x, dadj, dyn, Wn, bn, dims, num_route, HS, HT):
    """x (BTN,C); dadj (BT,HS*N), dyn (B,HT,T*HS), Wn (N,C,C), bn (N,C) precomputed -> out, c (BT,HS,N), saved."""
    B, T, N, C = dims
    rec, c, (s, v, Ht, Rt, Y) = cap_head_fwd(p, pfx, x, dadj, dyn, dims, num_route, HS, HT)
    out = ops.apply(rec, Wn, MODE_NODE, B * T, N, bias=bn, resid=x, epi=EPI_RES_LRELU)                                # :139-141
    return out, c, (x, out, rec, c, s, v, Ht, Rt, dyn, Wn, Y)


# Forward chains on the (sample, 16-node) slab (r04, gptst_hypertem_chain_fwd): consecutive hyperTem layers are node-local, so [hyperTem2, hyperTem3]
# and [hyperTem4, the next STHCN's hyperTem1] run as ONE launch each, and the chained layer has no load phase of its own.  GPTST_CHAIN_FWD=0: one
# launch per layer.  (The caps' node layers stay on the node-grouped apply64 — it shares W_n over the 384 (b,t) rows of a node: 12.6 us against
# ~18 us for the per-sample form inside the chain launch, measured in r04 and removed in r05.)
CHAIN_FWD = os.environ.get("GPTST_CHAIN_FWD", "1") == "1"


def chain_fwd_ok(dims):
    return CHAIN_FWD and dims[3] == 64 and not DROP_R      # (node shards too: the chained layers are node-local, r05)


def ht_chain_fwd(x, stages, dims):
    """consecutive hyperTem layers in one launch -> [saved tuple per layer], last output.  stages: [(G, Wbt, bbt), ...]"""
    B, T, N, C = dims
    res = ops.hypertem_chain_fwd(x.view(B, T, N, C), stages)
    saved, xin = [], x
    for (G, Wbt, _b), (R, o) in zip(stages, res):
        o = o.view(-1, C)
        saved.append((xin, R.view(-1, C), o, G, Wbt))
        xin = o
    return saved, xin


def cap_core_bwd(p, g, pfx, saved, dout, dims, HS, HT, red, chain=False):
    """-> dx and the pieces whose reductions are batched by the caller: (dWn, nsplit, dbn, ddyn, dlogit).
    chain: dout already is dPre, and dx is returned multiplied by lrelu'(x) (x is a hyperTem output)."""
    B, T, N, C = dims
    x, out, rec, c, s, v, Ht, Rt, dyn, Wn, Y = saved
    BT, dev = B * T, x.device
    assert not chain or chain_ok(dims)
    if C == 64:     # data gradient, weight gradient and bias gradient of the node-conditioned layer in one pass
        drec, dWn, dbn, ns = ops.apply_wgrad(dout, None if chain else out, rec, Wn, MODE_NODE, BT, N)
        nsb = ns
    elif chain:     # C = 128, dPre chain: neither pass reads the layer's output
        drec = ops.apply(dout, Wn, MODE_NODE, BT, N, transw=True)
        dWb, ns = ops.wgrad(rec, dout, MODE_NODE, BT, N, colsum_d=True)                                          # rows [dWn | dbn] per split
        dWn, dbn, nsb = dWb[:, :C * C], dWb[:, C * C:], ns
    else:
        drec = ops.apply(dout, Wn, MODE_NODE, BT, N, A2=out, transw=True, pro=PRO_DPRE)
        dWb, ns = ops.wgrad(rec, dout, MODE_NODE, BT, N, D2=out, pro=PRO_DPRE, colsum_d=True)                    # rows [dWn | dbn] per split
        dWn, dbn, nsb = dWb[:, :C * C], dWb[:, C * C:], ns
    fused = None
    dc1, dv = ops.cap_rec_bwd(drec, c, v, reduce_nodes=CTX.NODE_REDUCE)
    gw, gb = g[pfx + "ln_p.weight"], g[pfx + "ln_p.bias"]
    if fused is None and FUSE_CROSS and CAP_LIN and C == 64 and CTX.NODE_REDUCE is None and Y is None:
        # r05: cross-time backward (role) + routing backward + the entry Linear's backward and the residual branch in ONE launch; dY never leaves LDS
        # r05: queued reductions of the layers already behind us ride in this launch — where it has the role form (idle slots: B*T + 4B <= 512 workgroups)
        carry = red.take_carry(CARRY_MB) if cross_role() and _C.lib().value("gptst_cap_route_roles_ok", B, T, N, C, HS, HT) == 1 else None
        lin = ops.cap_cross_route_lin_bwd(x.view(B, T, N, C), p[pfx + "ln_p.weight"], p[pfx + "ln_p.bias"], c, dc1, dv, s, Rt, Ht, dyn,
                                          p[pfx + "mask_template"], dout, None if chain else out, chain, B, T, HS, HT,
                                          flags=_zeros(x, 4 * B) if cross_role() else None, jobs=carry)
        red.untake(carry)                                              # (not launched: shape beyond the fused form)
        if lin is not None:
            dx, dWp, dbp, dlogit, ddyn = lin
            red.jobs.bwd_pool(_ones(dev, dWp.shape[0]), dWp, gw.view(1, C * C))      # B*T (+ the node halves' rows, r06) partials
            red.jobs.bwd_pool(_ones(dev, dbp.shape[0]), dbp, gb.view(1, C))
            return dx, (dWn, ns, (dbn, nsb), ddyn, dlogit)
    if fused is None and FUSE_CROSS and CTX.NODE_REDUCE is None and Y is None:
        fused = ops.cap_cross_route_bwd(x.view(B, T, N, C), p[pfx + "ln_p.weight"], p[pfx + "ln_p.bias"], c, dc1, dv, s, Rt, Ht, dyn,
                                        p[pfx + "mask_template"], B, T, HS, HT, flags=_zeros(x, 4 * B) if cross_role() else None)
    if fused is not None:
        dY, dlogit, ddyn = fused
    else:
        dS, ddyn = ops.cap_cross_bwd(dv, s, Rt, Ht, dyn, p[pfx + "mask_template"], B, T, HS, HT)
        dY, dlogit = ops.cap_route_bwd(x.view(B, T, N, C), p[pfx + "ln_p.weight"], p[pfx + "ln_p.bias"], c, dc1, dS, Y=Y)
    if C == 64:
        # dx = dY Wp + dout*lrelu'(out), the ln_p weight gradient and its bias gradient in one pass over dY
        dx, dWp, dbp, ns2 = ops.linear_bwd(dY, x, p[pfx + "ln_p.weight"], dout, None if chain else out, premul=chain)
        red.jobs.bwd_pool(_ones(dev, ns2), dWp, gw.view(1, C * C))
        red.jobs.bwd_pool(_ones(dev, ns2), dbp, gb.view(1, C))
    else:
        if chain:   # (dY Wp + dPre) * lrelu'(x): the sign comes from the layer's INPUT, which the weight gradient below reads anyway
            dx = ops.apply(dY, p[pfx + "ln_p.weight"], MODE_SHARED, BT, N, resid=dout, resid2=x, epi=EPI_ADD_PREMUL)
        else:
            dx = ops.apply(dY, p[pfx + "ln_p.weight"], MODE_SHARED, BT, N, resid=dout, resid2=out, epi=EPI_ADD_DPRE)
        dWp, ns2 = ops.wgrad(dY, x, MODE_SHARED, BT, N, colsum_a=True)                     # rows [dWp | colsum dY]
        red.jobs.bwd_pool(_ones(dev, ns2), dWp[:, :C * C], gw.view(1, C * C))
        red.jobs.bwd_pool(_ones(dev, ns2), dWp[:, C * C:], gb.view(1, C))
    return dx, (dWn, ns, (dbn, nsb), ddyn, dlogit)


_ONES = {}


def _ones(dev, n=1):
    """cached (n, 1) column of ones: the 'embedding' of a plain sum over n partial rows"""
    if (dev, n) not in _ONES:
        _ONES[(dev, n)] = torch.ones(n, 1, device=dev)
    return _ONES[(dev, n)]


# ---- LReLU(x W_g + b_g) with generated weights, no residual (MLP_RL, GPTST.py:24-32) --------------------------------
def condlin_fwd(x, Wg, bg, mode, dims):
    B, T, N, C = dims
    out = ops.apply(x, Wg, mode, B * T, N, bias=bg, epi=EPI_LRELU)
    return out, (x, out, Wg)


def condlin_bwd(saved, dout, emb, wpool, bpool, g_wpool, g_bpool, d_emb, mode, dims, red, chain=False, premul=False):
    """chain: dout already is dPre;  premul (chain only): dx is returned multiplied by lrelu'(x)."""
    B, T, N, C = dims
    x, out, Wg = saved
    R, K = emb.shape
    assert not chain or chain_ok(dims)
    if C == 64:
        dx, dW, db, ns = ops.apply_wgrad(dout, None if chain else out, x, Wg, mode, B * T, N, premul=chain and premul)
        nsb = ns
    elif chain:     # C = 128, dPre chain
        dx = (ops.apply(dout, Wg, mode, B * T, N, transw=True, resid2=x, epi=EPI_PREMUL) if premul
              else ops.apply(dout, Wg, mode, B * T, N, transw=True))
        dWb, ns = ops.wgrad(x, dout, mode, B * T, N, colsum_d=True)
        dW, db, nsb = dWb[:, :C * C], dWb[:, C * C:], ns
    else:
        dx = ops.apply(dout, Wg, mode, B * T, N, A2=out, transw=True, pro=PRO_DPRE)
        dWb, ns = ops.wgrad(x, dout, mode, B * T, N, D2=out, pro=PRO_DPRE, colsum_d=True)
        dW, db, nsb = dWb[:, :C * C], dWb[:, C * C:], ns
    dW = dW if dW.dim() == 2 else dW.view(ns * R, C * C)
    red.jobs.bwd_pool(emb, dW, g_wpool.view(K, C * C), nsplit=ns)
    red.jobs.bwd_pool(emb, db, g_bpool, nsplit=nsb)
    red.jobs.bwd_emb(dW, wpool.view(K, C * C), d_emb, nsplit=ns)
    red.jobs.bwd_emb(db, bpool, d_emb, nsplit=nsb)
    return dx


# ---- STHCN (GPTST.py:253-273) --------------------------------------------------------------------------------------
ENC, DEC = "encoder.STHCN_encode.", "decoder.STHCN_decode."
GUIDE_TF = "encoder.teb4mask."


def _sthcn_names(pfx):
    return [pfx + "hyperTem%d." % i for i in (1, 2, 3, 4)], [pfx + "cap1.", pfx + "cap2."]


def _sthcn_gen_jobs(p, pfx, emb, jobs, A_all, dims, G_all=None, gjobs=None):
    """Queue the generated-parameter problems of one STHCN (20 jobs); -> gen dict (tensors are filled by jobs.launch()).  gjobs: the table the
    temporal-graph jobs go to (default: jobs)."""
    gjobs = jobs if gjobs is None else gjobs
    B, T, N, C = dims
    time_eb, teb, tes = emb
    ne, nes = p[pfx + "node_embeddings"], p[pfx + "node_embeddings_spg"]
    hts, cps = _sthcn_names(pfx)
    adj0 = p[hts[0] + "adj"]
    d, Hm = adj0.shape[0], adj0.shape[1]
    cadj, tadj = p[cps[0] + "adj"], p[cps[0] + "t_adj"]
    ds, HS, HT = cadj.shape[0], cadj.shape[1], tadj.shape[1]
    for i, h in enumerate(hts):
        jobs.fwd(ne, p[h + "adj"].view(d, Hm * T), out=A_all[i])                                          # :156
        if G_all is not None:
            gjobs.gram(ne, p[h + "adj"].view(d, Hm * T), out=G_all[i], A=A_all[i])                        # :156-158 G_n = A_n^T A_n, same launch
    Wb = [jobs.fwd(time_eb, t) for h in hts for t in (p[h + "weights_pool"], p[h + "bias_pool"])]         # :160-161
    Wn = [jobs.fwd(nes, t) for c in cps for t in (p[c + "weights_spa"], p[c + "bias_spa"])]               # :137-138
    dadj = [jobs.fwd(teb, p[c + "adj"].view(ds, HS * N)) for c in cps]                                    # :104
    dyn = [jobs.fwd(tes, p[c + "t_adj"].view(ds, HT * T * HS)).view(B, HT, T * HS) for c in cps]          # :129
    return dict(emb=emb, gen=(A_all, hts, cps, d, Hm, ds, HS, HT), Wb=Wb, Wn=Wn, dadj=dadj, dyn=dyn)


def gen_all(p, tidx, dims, which=(ENC, DEC), guide=True, defer=False):
    """Everything of a step that depends only on the time index and the parameters — the seven time embeddings (:256-261, :337)
    and every generated parameter of both STHCNs and of the guide MLP — in THREE launches (one time-feature job table, one
    poolgen job table, one gram) instead of 23.  -> {prefix: gen dict, "guide": (t4m, Wspa, bspa, Wtem, btem)}
    defer (r05): only the guide's parameters are generated here; the jobs of the STHCNs (parameters and temporal graphs) come back unlaunched as
    res["pending"] (a PoolJobs) — the stepper hands them to the mask generation, whose cooperative launch runs them on the CUs it leaves idle
    (ops.mask_random / mask_adaptive jobs=); nothing before the mask reads their outputs."""
    B, T, N, C = dims
    tfj = []
    for pfx in which:
        tfj += [_tf_job(p, pfx + "time_feature1.", tidx), _tf_job(p, pfx + "time_feature1_.", tidx),
                _tf_job(p, pfx + "time_feature2.", tidx, spg=True)]
    if guide:
        tfj.append(_tf_job(p, GUIDE_TF, tidx))
    embs = ops.timefeat_jobs_fwd(tfj, tidx)
    jobs = ops.PoolJobs()
    now = ops.PoolJobs() if defer else jobs
    res = {}
    L = 4 * len(which)
    if guide:
        gj = now
        m = "encoder.MLP_RL."
        t4m = embs[-1]
        res["guide"] = (t4m, gj.fwd(p["encoder.neb4mask"], p[m + "weights_pool_spa"]), gj.fwd(p["encoder.neb4mask"], p[m + "bias_pool_spa"]),
                        gj.fwd(t4m, p[m + "weights_pool_tem"]), gj.fwd(t4m, p[m + "bias_pool_tem"]))
    if L:
        adj0 = p[which[0] + "hyperTem1.adj"]
        Hm = adj0.shape[1]
        A_all = torch.empty(L, N, Hm * T, device=tidx.device)
        G_all = torch.empty(L, N, T, T, device=tidx.device)
        for k, pfx in enumerate(which):
            res[pfx] = _sthcn_gen_jobs(p, pfx, embs[3 * k:3 * k + 3], jobs, A_all[4 * k:4 * k + 4], dims, G_all[4 * k:4 * k + 4])
            res[pfx]["slot"] = (k, len(which))
            res[pfx]["G_all"] = G_all[4 * k:4 * k + 4]

    if defer:                                            # (a temporal graph beyond the job kernel's shapes: jobs.post — the table then launches on its own)
        now.launch()
        res["pending"] = jobs
        return res
    jobs.launch()                                        # generated parameters AND the temporal graphs: one launch
    return res


def sthcn_fwd(p, pfx, tidx, x, dims, num_route, gen=None, head=None, next_gen=None):
    """head: (x after hyperTem1, its saved tuple) when the previous STHCN's last chain already ran this one's first layer;
    next_gen: gen dict of the NEXT STHCN — its hyperTem1 then rides on this one's last chain; -> x, c1, saved[, next head]."""
    if gen is None:
        gen = gen_all(p, tidx, dims, which=(pfx,), guide=False)[pfx]
    A_all, hts, cps, d, Hm, ds, HS, HT = gen["gen"]
    G_all, Wb, Wn, dadj, dyn = gen["G_all"], gen["Wb"], gen["Wn"], gen["dadj"], gen["dyn"]
    sv = {}
    nhead = None
    if head is not None:
        x, sv["h1"] = head
    else:
        x, sv["h1"] = hypertem_core_fwd(x, G_all[0], Wb[0], Wb[1], dims)
    if chain_fwd_ok(dims):
        # hyperTem PAIRS on the slab (the caps' node layers stay on the node-grouped apply64)
        x, c1, sv["c1"] = cap_core_fwd(p, cps[0], x, dadj[0], dyn[0], Wn[0], Wn[1], dims, num_route, HS, HT)
        (sv["h2"], sv["h3"]), x = ht_chain_fwd(x, [(G_all[1], Wb[2], Wb[3]), (G_all[2], Wb[4], Wb[5])], dims)
        x, _, sv["c2"] = cap_core_fwd(p, cps[1], x, dadj[1], dyn[1], Wn[2], Wn[3], dims, num_route, HS, HT)
        if next_gen is not None:
            hs, xl = ht_chain_fwd(x, [(G_all[3], Wb[6], Wb[7]), (next_gen["G_all"][0], next_gen["Wb"][0], next_gen["Wb"][1])], dims)
            sv["h4"] = hs[0]
            x = hs[0][2]
            nhead = (xl, hs[1])
        else:
            x, sv["h4"] = hypertem_core_fwd(x, G_all[3], Wb[6], Wb[7], dims)
    else:
        x, c1, sv["c1"] = cap_core_fwd(p, cps[0], x, dadj[0], dyn[0], Wn[0], Wn[1], dims, num_route, HS, HT)
        x, sv["h2"] = hypertem_core_fwd(x, G_all[1], Wb[2], Wb[3], dims)
        x, sv["h3"] = hypertem_core_fwd(x, G_all[2], Wb[4], Wb[5], dims)
        x, _, sv["c2"] = cap_core_fwd(p, cps[1], x, dadj[1], dyn[1], Wn[2], Wn[3], dims, num_route, HS, HT)
        x, sv["h4"] = hypertem_core_fwd(x, G_all[3], Wb[6], Wb[7], dims)
    sv["emb"] = gen["emb"]
    sv["gen"] = gen["gen"]
    sv["slot"] = gen.get("slot")
    if next_gen is not None:
        return x, c1, sv, nhead
    return x, c1, sv


def _grad_buffers(red, slot, N, T, HmT, ref, nsG):
    """dG (4, nsG, N,T,T) partial graph gradients and dA (4,N,Hm*T) of one STHCN.  Both STHCNs of a step get adjacent halves of one
    buffer each, so that Reductions.flush() turns their temporal-graph gradients into the factor gradients with ONE gram_bwd launch."""
    if slot is None:
        return torch.empty(4, nsG, N, T, T, device=ref.device), torch.empty(4, N, HmT, device=ref.device)
    k, n = slot
    if getattr(red, "_dG", None) is None or red._dG.shape[0] != 4 * n:
        red._dG = torch.empty(4 * n, nsG, N, T, T, device=ref.device)
        red._dA = torch.empty(4 * n, N, HmT, device=ref.device)
    return red._dG[4 * k:4 * k + 4], red._dA[4 * k:4 * k + 4]


def sthcn_bwd(p, g, pfx, tidx, sv, dout, dims, red, chain=False, premul_in=False, defer_h1=False, after_pending=None):
    """chain: dout already is dPre of the last layer and every layer hands dPre down;  premul_in (chain only): the returned input gradient
    is multiplied by lrelu'(input) — True when the STHCN's input is itself a LeakyReLU output (the decoder's: the encoder embedding).
    defer_h1 (chain, premul_in): hyperTem1's backward is NOT run — a PendingH1 is returned and the STHCN below runs it in the pair launch
    with its own hyperTem4 (dout: that PendingH1)."""
    B, T, N, C = dims
    time_eb, teb, tes = sv["emb"]
    A_all, hts, cps, d, Hm, ds, HS, HT = sv["gen"]
    ne, nes = p[pfx + "node_embeddings"], p[pfx + "node_embeddings_spg"]
    dne, dnes = g[pfx + "node_embeddings"], g[pfx + "node_embeddings_spg"]
    d_te, d_teb, d_tes = _zeros(time_eb, *time_eb.shape), _zeros(teb, *teb.shape), _zeros(tes, *tes.shape)
    nsG = graph_grad_splits(dims)
    dG_all, dA_all = _grad_buffers(red, sv.get("slot"), N, T, Hm * T, dout, nsG)
    # ---- gradient reductions of the generated parameters: queued as soon as a layer's partials have been launched (r05: the routing backward launches
    # further down the chain carry them as role workgroups, red.take_carry), executed by red.flush() at the latest ----
    J = red.jobs
    CC, BT = C * C, B * T

    def queue_ht(h, hp):
        dWbt, ns, (dbias, nsb) = hp                              # (ns*BT, CC), possibly a column window of [dW | db] rows
        J.bwd_pool(time_eb, dWbt, g[h + "weights_pool"].view(d, CC), nsplit=ns)
        J.bwd_pool(time_eb, dbias, g[h + "bias_pool"], nsplit=nsb)
        J.bwd_emb(dWbt, p[h + "weights_pool"].view(d, CC), d_te, nsplit=ns)
        J.bwd_emb(dbias, p[h + "bias_pool"], d_te, nsplit=nsb)

    def queue_cap(c, cp):
        dWn, ns, (dbn, nsb), ddyn, dlogit = cp
        dW2 = dWn if dWn.dim() == 2 else dWn.view(ns * N, CC)      # (ns*N, CC), possibly a column window of [dW | db] rows
        J.bwd_pool(nes, dW2, g[c + "weights_spa"].view(d, CC), nsplit=ns)
        J.bwd_pool(nes, dbn, g[c + "bias_spa"], nsplit=nsb)
        J.bwd_emb(dW2, p[c + "weights_spa"].view(d, CC), dnes, nsplit=ns)
        J.bwd_emb(dbn, p[c + "bias_spa"], dnes, nsplit=nsb)
        dd2 = ddyn.view(B, HT * T * HS)
        J.bwd_pool(tes, dd2, g[c + "t_adj"].view(ds, HT * T * HS))
        J.bwd_emb(dd2, p[c + "t_adj"].view(ds, HT * T * HS), d_tes)
        dl2 = dlogit.view(BT, HS * N)
        J.bwd_pool(teb, dl2, g[c + "adj"].view(ds, HS * N))
        J.bwd_emb(dl2, p[c + "adj"].view(ds, HS * N), d_teb)

    if isinstance(dout, PendingH1):                 # the STHCN above left its first layer to the pair launch with this one's last layer
        up = dout
        dd, _, hp4 = ht_pair_bwd(up.saved, sv["h4"], up.dd, up.dG, dG_all[3], dims, dWb1=up.dWb)
        red.keep.append(up)
        if after_pending is not None:               # the STHCN above is complete only now (data parallel: its gradient bucket closes here)
            after_pending()
    else:
        dd, hp4 = hypertem_core_bwd(sv["h4"], dout, dG_all[3], dims, chain, True)
    queue_ht(hts[3], hp4)
    dd, cp2 = cap_core_bwd(p, g, cps[1], sv["c2"], dd, dims, HS, HT, red, chain)
    queue_cap(cps[1], cp2)
    pair = ht_pair_bwd(sv["h3"], sv["h2"], dd, dG_all[2], dG_all[1], dims) if chain and pair_bwd_on() else None
    if pair is not None:                            # hyperTem3 + hyperTem2: nothing in between (GPTST.py:267-268) -> one launch on the slab
        dd, hp3, hp2 = pair
    else:
        dd, hp3 = hypertem_core_bwd(sv["h3"], dd, dG_all[2], dims, chain, True)
        dd, hp2 = hypertem_core_bwd(sv["h2"], dd, dG_all[1], dims, chain, True)
    queue_ht(hts[2], hp3)
    queue_ht(hts[1], hp2)
    dd, cp1 = cap_core_bwd(p, g, cps[0], sv["c1"], dd, dims, HS, HT, red, chain)
    queue_cap(cps[0], cp1)
    if isinstance(sv["h1"], EncIn):                 # the encoder's first layer on the low-rank input form: no input gradient tensor
        assert chain
        e = sv["h1"]
        w, bi = p["encoder.dim_in_flow.weight"], p["encoder.dim_in_flow.bias"]
        if nsG == B:
            dWb, _, dinp = ops.encin_ht1_bwd(dd.view(B, T, N, C), e.source, e.mask, e.fill, w, bi, e.Wbt, e.ab, e.wv, dG=dG_all[0])
        else:       # (C = 128: the other layers write ONE graph-gradient partial; this kernel writes one per sample)
            dWb, dGb, dinp = ops.encin_ht1_bwd(dd.view(B, T, N, C), e.source, e.mask, e.fill, w, bi, e.Wbt, e.ab, e.wv)
            torch.sum(dGb, 0, out=dG_all[0][0])
        hp1 = (dWb[:, :C * C], 1, (dWb[:, C * C:], 1))
        wb = _wb_view(g["encoder.dim_in_flow.weight"], g["encoder.dim_in_flow.bias"])
        if wb is not None:
            red.jobs.bwd_pool(_ones(dd.device, dinp.shape[0]), dinp, wb)
        else:
            g["encoder.dim_in_flow.weight"].add_(dinp[:, :C].sum(0).view(C, 1))
            g["encoder.dim_in_flow.bias"].add_(dinp[:, C:].sum(0))
        red.keep.append((dinp, dWb))
        dd = None
    elif defer_h1:
        assert chain and premul_in
        ns = ops.wgrad_nsplit(MODE_TIME, B * T, N, C)
        dWb = torch.empty(ns * B * T, C * C + C, device=dd.device, dtype=torch.float32)
        hp1 = (dWb[:, :C * C], ns, (dWb[:, C * C:], ns))
        dd = PendingH1(sv["h1"], dd, dG_all[0], dWb)
    else:
        dd, hp1 = hypertem_core_bwd(sv["h1"], dd, dG_all[0], dims, chain, premul_in)
    queue_ht(hts[0], hp1)                           # (a deferred hyperTem1: its partials are written by the pair launch that opens the STHCN below)
    red.gram(A_all.view(4 * N, Hm, T), dG_all, dA_all.view(4 * N, Hm, T), N, nsG)
    for i, h in enumerate(hts):                                  # (dA_all is written by gram_bwd at flush time: the late table)
        red.late.bwd_pool(ne, dA_all[i], g[h + "adj"].view(d, Hm * T))
        red.late.bwd_emb(dA_all[i], p[h + "adj"].view(d, Hm * T), dne)
    red.timefeat(p, g, pfx + "time_feature1.", tidx, d_te)
    red.timefeat(p, g, pfx + "time_feature1_.", tidx, d_teb)
    red.timefeat(p, g, pfx + "time_feature2.", tidx, d_tes, spg=True)
    red.keep.append((sv, hp1, hp2, hp3, hp4, cp1, cp2, dout))
    return dd


# ---- whole model ---------------------------------------------------------------------------------------------------
class EncIn:
    """saved state of the encoder's first layer when it ran on the low-rank input form (ops.encin_ht1_fwd) instead of lin_in + hyperTem"""

    def __init__(self, source, mask, fill, Wbt, ab, wv):
        self.source, self.mask, self.fill, self.Wbt, self.ab, self.wv = source, mask, fill, Wbt, ab, wv


ENCIN = os.environ.get("GPTST_ENCIN", "1") == "1"


def encin_ok(dims, base):
    """input projection + encoder hyperTem1 as one rank-2 kernel pair (encin.hip): base = 1 and the dPre chain in the backward"""
    return ENCIN and base == 1 and dims[3] in (64, 128) and dims[1] == 12 and chain_ok(dims)       # node-local: serves node shards too (r05)


GUIDEIN = os.environ.get("GPTST_GUIDEIN", "1") == "1"
GUIDE_HEAD = os.environ.get("GPTST_GUIDE_HEAD", "1") == "1"     # r06: the classifier's forward as node vectors + ONE (b,t)-grouped pass (gptst_guide_head_fwd)


def guide_fwd(p, source, tidx, dims, base, gen=None, lowrank_in=False):
    """softmax(MLP_RL(raw flow, teb4mask(t), neb4mask)) — GPTST.py:326-332 / 337-343.  -> prob (BTN,HS), saved.
    lowrank_in (the caller's backward is the dPre chain): input projection + node-conditioned layer as the elementwise low-rank form of
    guidein.hip (base = 1): saved[1] is then the tuple ("lowrank", Wspa) instead of the layer's (x, out, W)."""
    B, T, N, C = dims
    if gen is None:
        gen = gen_all(p, tidx, dims, which=(), guide=True)["guide"]
    t4m, Wspa, bspa, Wtem, btem = gen
    m = "encoder.MLP_RL."
    if lowrank_in and GUIDEIN and GUIDE_HEAD and base == 1 and C == 64:
        # r06: u_n / c_n, then ONE (b,t)-grouped pass for :22-:33, the softmax and the labels (bit-identical to the three launches below)
        r = ops.guide_head_fwd(source, p[m + "ln1.weight"], p[m + "ln1.bias"], Wspa, bspa, Wtem, btem, p[m + "ln3.weight"], p[m + "ln3.bias"])
        if r is not None:
            h1, h2, prob, label = r
            return prob, (t4m, ("lowrank", Wspa), (h1, h2, Wtem), h2, label)
    if lowrank_in and GUIDEIN and base == 1 and C in (64, 128):
        h1 = ops.guide_in_fwd(source, p[m + "ln1.weight"], p[m + "ln1.bias"], Wspa, bspa)                 # :22 + :24-27, elementwise
        s1 = ("lowrank", Wspa)
    else:
        h0 = ops.lin_in(source, base + 2, base, p[m + "ln1.weight"], p[m + "ln1.bias"], C)                # :22
        h1, s1 = condlin_fwd(h0, Wspa, bspa, MODE_NODE, dims)                                             # :24-27
    h2, s2 = condlin_fwd(h1, Wtem, btem, MODE_TIME, dims)                                                 # :29-32
    prob, label = ops.rowdot(h2, p[m + "ln3.weight"], p[m + "ln3.bias"], softmax=True, want_label=True)   # :33, :332, :344-345
    return prob, (t4m, s1, s2, h2, label)


def _wb_view(gw, gb):
    """[weight | bias] gradient of an nn.Linear as ONE (1, J*C + J) row when the two tensors are adjacent in the flat buffer."""
    n = gw.numel() + gb.numel()
    if gb.data_ptr() != gw.data_ptr() + 4 * gw.numel():
        return None
    return torch.as_strided(gw, (1, n), (n, 1))


def fused_tails_ok(p, C, base, HS):
    """The fused loss heads (tails.hip) serve C in {64, 128} and J, HS <= 16 with [weight | bias] adjacent in the flat buffer."""
    return (C in (64, 128) and base <= ops.TAIL_MAXJ and HS <= ops.TAIL_MAXJ
            and _wb_view(p["decoder.dim_flow_out.weight"], p["decoder.dim_flow_out.bias"]) is not None
            and _wb_view(p["encoder.MLP_RL.ln3.weight"], p["encoder.MLP_RL.ln3.bias"]) is not None)


def loss_tail(p, g, dec, source, mask, base, sigma, mu, thresh, sws, red, chain=False):
    """dim_flow_out + masked MAE + their backward in one pass over dec (GPTST.py:455, Run.py:92-100) -> out (BTN, base), d_dec.
    sws: the step's loss-statistics scratch (ops.tail_sws / arena zeros), folded into stats by ops.stats_fold.
    chain: d_dec is returned multiplied by lrelu'(dec) (the dPre of the decoder's last hyperTem layer)."""
    wo = "decoder.dim_flow_out."
    out, dd, part = ops.tail_mae(dec, p[wo + "weight"], p[wo + "bias"], source, base + 2, mask, sigma, mu, thresh, sws, premul=chain)
    red.jobs.bwd_pool(_ones(dec.device, part.shape[0]), part, _wb_view(g[wo + "weight"], g[wo + "bias"]))
    red.keep.append((part, dd))
    return out, dd


def kl_head(p, g, sv_g, prob, c1, N, w, sws, red, chain=False):
    """0.1 KL(eb || prob) and the backward through softmax + MLP_RL.ln3 in one pass over h2 -> d_h2 (guide_bwd(dh2=...));
    chain: multiplied by lrelu'(h2)."""
    m = "encoder.MLP_RL."
    dh2, part = ops.tail_kl(sv_g[3], p[m + "ln3.weight"], prob, c1, N, w, sws, premul=chain)
    red.jobs.bwd_pool(_ones(prob.device, part.shape[0]), part, _wb_view(g[m + "ln3.weight"], g[m + "ln3.bias"]))
    red.keep.append((part, dh2))
    return dh2


def guide_bwd(p, g, source, tidx, saved, dlogit, dims, base, red, dh2=None, chain=False):
    """dlogit (BTN,HS): gradient of the logits — or dh2 (BTN,C) when kl_head already went through ln3 (chain: dh2 is dPre)."""
    B, T, N, C = dims
    t4m, s1, s2, h2 = saved[:4]
    m = "encoder.MLP_RL."
    if dh2 is None:
        HS = dlogit.shape[1]
        dh2 = ops.lin_in(dlogit, HS, HS, p[m + "ln3.weight"], None, C, wlayout=1)
        ops.rowouter(dlogit, HS, HS, h2, g[m + "ln3.weight"], 1, asum=g[m + "ln3.bias"])
    d_t4m = _zeros(t4m, *t4m.shape)
    dh1 = condlin_bwd(s2, dh2, t4m, p[m + "weights_pool_tem"], p[m + "bias_pool_tem"], g[m + "weights_pool_tem"],
                      g[m + "bias_pool_tem"], d_t4m, MODE_TIME, dims, red, chain, True)
    if isinstance(s1[0], str):      # ("lowrank", Wspa): node layer + input projection on the low-rank form: p_n, q_n per node instead of dh0 / h0
        assert chain
        neb, wpool, bpool = p["encoder.neb4mask"], p[m + "weights_pool_spa"], p[m + "bias_pool_spa"]
        K = neb.shape[1]
        dWb, dinp = ops.guide_in_bwd(dh1, source, p[m + "ln1.weight"], p[m + "ln1.bias"], s1[1])
        dW, db = dWb[:, :C * C], dWb[:, C * C:]
        red.jobs.bwd_pool(neb, dW, g[m + "weights_pool_spa"].view(K, C * C))
        red.jobs.bwd_pool(neb, db, g[m + "bias_pool_spa"])
        red.jobs.bwd_emb(dW, wpool.view(K, C * C), g["encoder.neb4mask"])
        red.jobs.bwd_emb(db, bpool, g["encoder.neb4mask"])
        wb = _wb_view(g[m + "ln1.weight"], g[m + "ln1.bias"])
        if wb is not None:
            red.jobs.bwd_pool(_ones(dh1.device, N), dinp, wb)
        else:
            g[m + "ln1.weight"].add_(dinp[:, :C].sum(0).view(C, 1))
            g[m + "ln1.bias"].add_(dinp[:, C:].sum(0))
        red.keep.append((dWb, dinp))
        dh0 = None
    else:
        dh0 = condlin_bwd(s1, dh1, p["encoder.neb4mask"], p[m + "weights_pool_spa"], p[m + "bias_pool_spa"],
                          g[m + "weights_pool_spa"], g[m + "bias_pool_spa"], g["encoder.neb4mask"], MODE_NODE, dims, red, chain, False)
        _in_proj_grads(source, base, dh0, g[m + "ln1.weight"], g[m + "ln1.bias"], None, 0.0, red)
    red.timefeat(p, g, GUIDE_TF, tidx, d_t4m)
    red.keep.append((saved, dlogit, dh2, dh1, dh0))


def _in_proj_grads(source, base, dY, gW, gb, mask, fill, red):
    """Weight / bias gradient of an input projection Linear(base -> C) on the (masked) raw flow (GPTST.py:22, :416-418): gW (C, base) +=
    dY^T a', gb += colsum(dY).  base = 1: the row-chunk partials [gW | gb] fold as ONE job of the step's reduction launch; else the
    two-launch rowouter."""
    wb = _wb_view(gW, gb)
    if base == 1 and wb is not None:
        C = dY.shape[1]
        part = ops.rowouter_part(source, base + 2, base, dY, mask=mask, fill=fill)
        red.jobs.bwd_pool(_ones(dY.device, part.shape[0]), part[:, :2 * C], wb)
        red.keep.append(part)
    else:
        ops.rowouter(source, base + 2, base, dY, gW, 0, csum=gb, mask=mask, fill=fill)


def model_fwd(p, source, mask, dims, base, num_route, scaler_zeros, gen=None, tidx=None, dec_gen=None, lowrank_in=False):
    """Masked-autoencoder body — GPTST.py:415-421 + 453-456.  mask (BTN*base) fp32, 1 = visible; None -> no masking (eval).
    dec_gen: the decoder STHCN's gen dict — its first hyperTem layer then rides on the encoder's last chain launch and the result comes back
    as a fifth value, to be passed to decoder_fwd(dec_head=...) (None when the chain path does not serve the shape).
    lowrank_in: the caller's backward is the dPre chain (model_bwd(chain=True)) — the input projection + encoder hyperTem1 then run as the
    rank-2 kernel pair of encin.hip where the shape allows."""
    B, T, N, C = dims
    if tidx is None:
        tidx = source[:, :, 0, base:base + 2].contiguous()
    head = None
    if lowrank_in and gen is not None and encin_ok(dims, base):
        # input projection + encoder hyperTem1 on the rank-2 structure of the input: no x0, no GEMM (ops.encin_ht1_fwd)
        w, bi = p["encoder.dim_in_flow.weight"], p["encoder.dim_in_flow.bias"]
        G1, Wb = gen["G_all"][0], gen["Wb"]
        o1, ab, wv = ops.encin_ht1_fwd(source, base, mask, 0.0 if mask is None else scaler_zeros, w, bi, G1, Wb[0], Wb[1])
        head = (o1.view(-1, C), EncIn(source, mask, 0.0 if mask is None else scaler_zeros, Wb[0], ab, wv))
        x0 = None
    else:
        x0 = ops.lin_in(source, base + 2, base, p["encoder.dim_in_flow.weight"], p["encoder.dim_in_flow.bias"], C,
                        mask=mask, fill=scaler_zeros)                                                      # :416-418
    if dec_gen is not None:
        emb, c1, sv_e, dec_head = sthcn_fwd(p, ENC, tidx, x0, dims, num_route, gen=gen, next_gen=dec_gen, head=head)
        return emb, c1, tidx, sv_e, dec_head
    emb, c1, sv_e = sthcn_fwd(p, ENC, tidx, x0, dims, num_route, gen=gen, head=head)                       # :421
    return emb, c1, tidx, sv_e


def decoder_fwd(p, tidx, emb, dims, num_route, gen=None, head=True, dec_head=None):
    dec, _, sv_d = sthcn_fwd(p, DEC, tidx, emb, dims, num_route, gen=gen, head=dec_head)                   # :454
    if not head:
        return None, dec, sv_d
    out = ops.rowdot(dec, p["decoder.dim_flow_out.weight"], p["decoder.dim_flow_out.bias"])                 # :455
    return out, dec, sv_d


def model_bwd(p, g, source, mask, tidx, sv_e, sv_d, dec, d_out, d_dec, dims, base, scaler_zeros, red, dd=None, chain=False):
    """Backward of decoder_fwd . model_fwd given d_out (BTN, base) [and optional d_dec (BTN, C)] — or given dd, the gradient
    w.r.t. the decoder STHCN output, when loss_tail already went through dim_flow_out.  Parameter-gradient reductions are queued
    on ``red`` (Reductions): the caller runs red.flush(tidx) once the whole backward is enqueued.
    chain (only with dd from loss_tail(chain=True)): the dPre chain, see chain_ok()."""
    B, T, N, C = dims
    wo = "decoder.dim_flow_out."
    assert not chain or dd is not None
    if dd is None:
        dd = ops.lin_in(d_out, base, base, p[wo + "weight"], None, C, wlayout=1)
        if d_dec is not None:
            dd = dd + d_dec
        ops.rowouter(d_out, base, base, dec, g[wo + "weight"], 1, asum=g[wo + "bias"])
    # the decoder's hyperTem1 and the encoder's hyperTem4 are adjacent (GPTST.py:271 -> :454): one pair launch, unless the decoder's gradient
    # bucket must be complete when its backward ends (data-parallel overlap / a side stream flush the decoder's reductions right here)
    # r05: also under the data-parallel bucket overlap — the decoder's bucket then closes ONE LAUNCH later, behind the pair launch that finishes its
    # first layer (the bucket's reductions and its forked all-reduce still run under the rest of the encoder's backward)
    defer = (chain and (red.on_bucket is None or PAIR_UNDER_DP) and not isinstance(sv_d["h1"], EncIn)
             and ht_pair_ok(sv_d["h1"], sv_e["h4"], dims))
    late = defer and red.on_bucket is not None
    d_emb = sthcn_bwd(p, g, DEC, tidx, sv_d, dd, dims, red, chain, True, defer_h1=defer)     # the decoder's input is the encoder's last LeakyReLU output
    if not late:
        red.flush_async(tidx)                               # the decoder's reductions overlap with the encoder's backward chain
    d_x0 = sthcn_bwd(p, g, ENC, tidx, sv_e, d_emb, dims, red, chain, False,  # the encoder's input is a plain Linear: no premultiplication
                     after_pending=(lambda: red.flush_async(tidx)) if late else None)
    red.flush_async(tidx)                                   # ... and the encoder's with the guide's
    if d_x0 is not None:                                    # (None: the low-rank first layer produced the input-projection gradient itself)
        _in_proj_grads(source, base, d_x0, g["encoder.dim_in_flow.weight"], g["encoder.dim_in_flow.bias"], mask, scaler_zeros, red)
