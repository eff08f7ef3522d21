"""Node-sharded pretraining step (BASELINE config 5: N = 4096 split 512 nodes per GPU).

Every rank owns a contiguous range of nodes [n0, n1): its slice of the input, of the node-indexed parameters
(``*.node_embeddings``, ``*.node_embeddings_spg``, ``encoder.neb4mask``, ``*.cap{1,2}.adj``) and of every activation.
hyperTem, MLP_RL, the in/out projections and the per-node parts of ``cap`` are node-local.  The only sums over nodes are the
cluster aggregations ``S = c . P`` of ``cap`` (R+2 = 4 per forward, 1 per backward: ``dv = sum_n c drec``); each is ONE kernel
(ops._capbig_type1) followed by ONE all-reduce of (B*T, HS, C) floats.  The cross-time hyperedge block of ``cap`` runs replicated
on the all-reduced ``s``, so the gradients it produces (``t_adj``, ``time_feature2``) are identical on every rank and are scaled
by 1/world before the gradient all-reduce.  Gradients of shared parameters: one all-reduce of the flat buffer (node-local slices
are put back afterwards — they belong to this rank alone); the global gradient norm adds the other ranks' node-local squared
norms (one scalar all-reduce).  Masks: the selection runs replicated over the GLOBAL (B,T,N) cells from identical noise and each
rank keeps its node columns; the adaptive phase all-gathers the per-cell cluster labels and sums the class counts.
Loss statistics (sum |e|, kept count, KL sum) travel in the tail of the gradient buffer, as in dist.py.

The collectives go through a small group object: ``DistNodeGroup`` (torch.distributed, RCCL on GPUs), ``NativeNodeGroup`` (the C-ABI
communicator of csrc/comm.hip: plain enqueues on the launch stream) or ``ThreadNodeGroup`` (ranks emulated by threads on ONE GPU —
how the tests check the protocol against the unsharded step).  A group whose collectives are stream-ordered device work
(``capturable``: world = 1, or NativeNodeGroup) lets the whole step — kernels AND collectives — be captured in ONE hipGraph per phase
(``use_graph``); otherwise the step is enqueued eagerly, the collectives sitting between the kernels.
"""
import os
import threading

import torch

from . import engine, ops
from .step import PretrainStep, U24, DEFER_GEN


def is_node_local(key):
    return (key.endswith("node_embeddings") or key.endswith("node_embeddings_spg") or key == "encoder.neb4mask"
            or (key.endswith(".adj") and ".cap" in key))


def is_replicated_compute(key):
    """Parameters whose gradient is computed identically on every rank (cross-time block on the all-reduced cluster capsules)."""
    return key.endswith(".t_adj") or ".time_feature2." in key


def shard_state_dict(sd, n0, n1):
    """Global state dict -> this rank's (node-indexed tensors sliced to [n0, n1))."""
    out = {}
    for k, v in sd.items():
        if is_node_local(k):
            out[k] = (v[..., n0:n1] if k.endswith(".adj") else v[n0:n1]).clone()
        else:
            out[k] = v.clone()
    return out


def unshard_state_dicts(sds):
    """Inverse of shard_state_dict for a list of per-rank state dicts (rank order = node order)."""
    out = {}
    for k, v in sds[0].items():
        if is_node_local(k):
            out[k] = torch.cat([sd[k] for sd in sds], dim=-1 if k.endswith(".adj") else 0)
        else:
            out[k] = v.clone()
    return out


class DistNodeGroup:
    """Collectives of a node-sharded run over torch.distributed (nccl = RCCL over xGMI on the GPUs)."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.capturable = world == 1                     # one rank: the collectives are no-ops / device copies

    def all_reduce_(self, t):
        import torch.distributed as dist
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def all_gather(self, t):
        """-> (world, *t.shape)"""
        import torch.distributed as dist
        flat = t.contiguous().view(-1)
        out = torch.empty(self.world * flat.numel(), dtype=t.dtype, device=t.device)      # concatenation form (gloo and nccl)
        if self.world == 1:
            out.copy_(flat)
        else:
            dist.all_gather_into_tensor(out, flat)
        return out.view((self.world,) + tuple(t.shape))


class NativeNodeGroup:
    """The same collectives on the C-ABI communicator (dist.NativeComm: RCCL bound at run time, every call a plain enqueue on torch's
    current stream), so they can sit INSIDE a captured hipGraph.  The library reduces fp32 only: integer payloads (class counts, cluster
    labels) travel as exactly representable floats (< 2^24), and the all-gather is an all-reduce of a buffer in which every rank fills its
    own slot."""
    capturable = True

    def __init__(self, comm):
        self.comm, self.rank, self.world = comm, comm.rank, comm.world

    def all_reduce_(self, t):
        if t.dtype == torch.float32:
            if t.is_contiguous():
                self.comm.allreduce_(t)
            else:                                   # reduce a contiguous copy and write the sums back (an in-place reduce of the copy would be lost)
                f = t.contiguous()
                self.comm.allreduce_(f)
                t.copy_(f)
            return t
        f = t.to(torch.float32)
        self.comm.allreduce_(f)
        t.copy_(f.to(t.dtype))
        return t

    def all_gather(self, t):
        """-> (world, *t.shape)"""
        buf = torch.zeros((self.world,) + tuple(t.shape), dtype=torch.float32, device=t.device)
        buf[self.rank].copy_(t)
        self.comm.allreduce_(buf)
        return buf.to(t.dtype)


class ThreadNodeGroup:
    """The same collectives between `world` threads of one process sharing one GPU stream (tests).  Kernels of all threads land
    on the same stream in enqueue order, and a barrier separates 'everybody has enqueued its contribution' from the sum."""

    capturable = False

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, rank, shared):
        self.rank, self.world, self.sh = rank, shared.world, shared

    def all_reduce_(self, t):
        sh = self.sh
        sh.slots[self.rank] = t
        sh.barrier.wait()
        total = sh.slots[0].clone()
        for r in range(1, self.world):
            total += sh.slots[r]
        sh.barrier.wait()                      # everybody has read every slot
        t.copy_(total)
        sh.barrier.wait()
        return t

    def all_gather(self, t):
        sh = self.sh
        sh.slots[self.rank] = t
        sh.barrier.wait()
        out = torch.stack([sh.slots[r] for r in range(self.world)])
        sh.barrier.wait()
        return out


class ShardedPretrainStep(PretrainStep):
    """One optimisation step of a rank that owns nodes [n0, n1) of N (all ranks: equal shard sizes)."""

    def __init__(self, model_local, args_local, n_global, group, scaler_mean, scaler_std, batch_size, seed=0, use_graph=None):
        """use_graph: None = capture the step in a hipGraph when the group's collectives are capturable (see the module docstring)."""
        super().__init__(model_local, args_local, scaler_mean, scaler_std, batch_size, use_graph=False, dp=None, seed=seed)
        self.shard_graph = bool(getattr(group, "capturable", False)) if use_graph is None else bool(use_graph)
        assert not self.shard_graph or getattr(group, "capturable", False), "this group's collectives cannot be captured"
        # r05: the node-local part of the step is the fused step's — loss / KL heads with their backward in one pass each (their per-workgroup
        # statistics are folded into the gradient buffer's tail BEFORE the all-reduce), the dPre chain, hyperTem forward chains and backward
        # pairs, the low-rank first layers.  Only the cluster aggregations of cap keep their all-reduce form (engine.CTX.NODE_REDUCE).
        self.fused_tails = (os.environ.get("GPTST_FUSED_TAILS", "1") == "1" and os.environ.get("GPTST_SHARD_FUSED", "1") == "1"
                            and engine.fused_tails_ok(model_local.param_views(), self.C, self.base, self.HS))
        self.group, self.Ng = group, n_global
        self.Nl = args_local.num_nodes
        assert self.Nl * group.world == n_global, "equal node shards"
        self.n0 = group.rank * self.Nl
        Mg = self.B * self.T * self.Ng
        torch.cuda.manual_seed(7654321 + seed)              # identical global mask noise on every rank
        self.noise_g = torch.zeros(Mg * self.base, device=self.dev)
        self.noise_ar_g = torch.zeros(2 * Mg, device=self.dev)      # adaptive phase: [noise_a | noise_r] over the GLOBAL cells, drawn by ONE launch
        self.noise_a_g, self.noise_r_g = self.noise_ar_g[:Mg], self.noise_ar_g[Mg:]
        self.tail = None                                    # single stream: collectives order against everything
        self.global_count_scale = True
        named = dict(model_local.named_parameters())
        self.local_keys = [k for k in named if is_node_local(k)]
        self.repl_keys = [k for k in named if is_replicated_compute(k)]
        self.segA = {k: model_local._offs[k] < model_local.nA for k in self.local_keys}
        # node-local gradients are kept out of the gradient all-reduce by a save / restore around it: ONE flat buffer, [reconstruction-path keys |
        # KL-path keys], moved by multi-tensor copies (r05: a clone, a copy and three norm launches PER KEY before — ~95 tiny launches per step)
        ordered = [k for k in self.local_keys if self.segA[k]] + [k for k in self.local_keys if not self.segA[k]]
        sizes = [named[k].numel() for k in ordered]
        self.keep_flat = torch.zeros(sum(sizes), device=self.dev)
        self.keep_nA = sum(n for k, n in zip(ordered, sizes) if self.segA[k])
        offs = [sum(sizes[:i]) for i in range(len(sizes))]
        self.keep_views = [self.keep_flat[o:o + n].view(named[k].shape) for k, o, n in zip(ordered, offs, sizes)]
        self.keep_grads = [self.g[k] for k in ordered]
        self.repl_grads = [self.g[k] for k in self.repl_keys]

    # global mask -> this rank's node columns
    def _cols(self, flat_global, per_cell):
        return flat_global.view(self.B, self.T, self.Ng, per_cell)[:, :, self.n0:self.n0 + self.Nl].contiguous().view(-1)

    def _mask(self, phase, prob, label=None, jobs=None):
        """label: the guide's argmax labels of the local cells (rowdot's by-product) — else taken from prob;  jobs: the STHCNs' generation table
        (engine.gen_all(defer=True)), launched by the mask call — inside its launch where that is the cooperative one"""
        a, base = self.args, self.base
        Mg = self.B * self.T * self.Ng
        ws = self.arena.zeros(ops.mask_ws_floats())        # the selections' histogram scratch: zeroed by the step's first launch
        if phase == 0:
            mask_g = ops.mask_random(self.noise_g, int(Mg * base * a.mask_ratio), ws=ws, u24=U24, jobs=jobs)
        else:
            if label is None:
                label = ops.mask_labels(prob)[0]                                               # local cells (B,T,Nl)
            if self.group.world == 1:
                label_g = label.view(-1)
            else:
                lab = self.group.all_gather(label.view(self.B, self.T, self.Nl))               # (W,B,T,Nl)
                label_g = lab.permute(1, 2, 0, 3).contiguous().view(-1)                        # (B,T,N) node-major within a cell row
            mask_g = ops.mask_adaptive(label_g, None, self.ctrl[:self.HS], self.ctrl[self.HS:], self.noise_a_g, self.noise_r_g,
                                       a.ada_type == "all", base, ws=ws, u24=U24, jobs=jobs)[2]  # (class histogram of the gathered labels: taken inside)
        self.last_mask_global = mask_g
        return self._cols(mask_g, base)

    def step(self, source, epoch, noise=None, noise_a=None, noise_r=None, list_c=None):
        """source: this rank's (B,T,Nl,base+2) slice; injected noise (tests) covers the GLOBAL (B,T,N[,base]) cells."""
        a = self.args
        phase = 0 if epoch <= a.change_epoch else 1
        if source is not self.src:
            self.src.copy_(source, non_blocking=True)
        inject = noise is not None or noise_a is not None
        if inject:
            if phase == 0:
                self.noise_g.copy_(noise.reshape(-1))
            else:
                self.noise_a_g.copy_(noise_a.reshape(-1)); self.noise_r_g.copy_(noise_r.reshape(-1))
        self._host_prepare(phase, epoch, list_c)
        #  what losses() needs to repeat this step after a lost hand-off: the base class's re-run calls step(src, epoch, list_c=...) — this one
        self._g_last = None
        self._last_call = (epoch, self._filled_list_c, self.rank_weight) if not inject else None
        self._unseen.append(phase)
        if len(self._unseen) > 4096:
            del self._unseen[:2048]
        if not self.shard_graph:
            self.inject_noise = inject
            self._sbody(phase)
            return
        key = (phase, inject)
        if key not in self.graphs:
            self._scapture(key)
        self.graphs[key].replay()

    def _scapture(self, key):
        """One hipGraph per (phase, injected noise): warm-up on a side stream, capture, undo the warm-up updates."""
        phase, self.inject_noise = key
        keep = (self.model.flat.clone(), self.m.clone(), self.v.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._sbody(phase)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._sbody(phase)
        self.graphs[key] = g
        self.model.flat.copy_(keep[0]); self.m.copy_(keep[1]); self.v.copy_(keep[2])
        torch.cuda.synchronize()

    def _sbody(self, phase):
        if not self.safe_mode:
            return self._sbody_impl(phase)
        with engine.no_handoffs():
            return self._sbody_impl(phase)

    def _sbody_impl(self, phase):
        """The device side of one step: kernels and collectives in stream order (eager, or inside a capture)."""
        mdl, a, base, dims = self.model, self.args, self.base, self.dims
        p, g = mdl.param_views(), self.g
        M = self.B * self.T * self.Nl
        ctx = engine.CTX
        ctx.ARENA, ctx.NODE_REDUCE = self.arena, self.group.all_reduce_
        try:
            src = self.src
            # zero_grad + the step's zero scratch + the time index (every node carries the same one, GPTST.py:256-257) + the GLOBAL mask noise
            # (Philox keyed by [seed, step]: identical on every rank) in ONE launch
            noise = None if self.inject_noise else (self.noise_g if phase == 0 else self.noise_ar_g)
            tidx = ops.step_begin(self.gbuf, self.arena.begin(zero=False), src, base, noise=noise, rng=self.rng_words)
            fused = self.fused_tails
            chain = fused and engine.chain_ok(dims)                # dPre chain: no backward kernel re-reads its layer's output
            need_guide = phase == 1 or not fused                   # (the fused form skips the classifier in the random-mask phase, as step.py does)
            gen = engine.gen_all(p, tidx, dims, guide=need_guide, defer=DEFER_GEN)
            red = engine.Reductions()
            prob, sv_g = engine.guide_fwd(p, src, tidx, dims, base, gen=gen["guide"], lowrank_in=chain) if need_guide else (None, None)
            mask = self._mask(phase, prob, sv_g[4] if sv_g is not None else None, jobs=gen.pop("pending", None))
            self.last_mask = mask
            if fused:
                dec_head = None
                if engine.chain_fwd_ok(dims):                      # the decoder's first hyperTem layer rides on the encoder's last chain launch
                    emb, c1, tidx, sv_e, dec_head = engine.model_fwd(p, src, mask, dims, base, mdl.num_route, mdl.scaler_zeros, gen=gen[engine.ENC],
                                                                     tidx=tidx, dec_gen=gen[engine.DEC], lowrank_in=chain)
                else:
                    emb, c1, tidx, sv_e = engine.model_fwd(p, src, mask, dims, base, mdl.num_route, mdl.scaler_zeros, gen=gen[engine.ENC], tidx=tidx,
                                                           lowrank_in=chain)
                _, dec, sv_d = engine.decoder_fwd(p, tidx, emb, dims, mdl.num_route, gen=gen[engine.DEC], head=False, dec_head=dec_head)
                sws = self.arena.zeros(ops.tail_parts(M), 4)       # per-workgroup loss statistics of the two heads
                out, dd = engine.loss_tail(p, g, dec, src, mask, base, self.std, self.mean, a.mape_thresh, sws, red, chain=chain)
                engine.model_bwd(p, g, src, mask, tidx, sv_e, sv_d, dec, None, None, dims, base, mdl.scaler_zeros, red, dd=dd, chain=chain)
                if phase == 1:
                    dh2 = engine.kl_head(p, g, sv_g, prob, c1, self.Nl, 0.1, sws, red, chain=chain)
                    engine.guide_bwd(p, g, src, tidx, sv_g, None, dims, base, red, dh2=dh2, chain=chain)
                ops.stats_fold(sws, self.stats)                    # ordered sum -> stats[0..2], inside the buffer the all-reduce below moves
            else:
                emb, c1, tidx, sv_e = engine.model_fwd(p, src, mask, dims, base, mdl.num_route, mdl.scaler_zeros, gen=gen[engine.ENC], tidx=tidx)
                out, dec, sv_d = engine.decoder_fwd(p, tidx, emb, dims, mdl.num_route, gen=gen[engine.DEC])
                ops.mae_fwd(out, src, base + 2, mask, self.std, self.mean, a.mape_thresh, M, base, self.stats)
                d_out = ops.mae_bwd(out, src, base + 2, mask, self.std, self.mean, a.mape_thresh, M, base, self.stats, normalize=False)
                engine.model_bwd(p, g, src, mask, tidx, sv_e, sv_d, dec, d_out, None, dims, base, mdl.scaler_zeros, red)
                if phase == 1:
                    dlogit = ops.kl(prob, c1, self.Nl, 0.1, self.stats)
                    engine.guide_bwd(p, g, src, tidx, sv_g, dlogit, dims, base, red)
            red.flush(tidx)
        finally:
            ctx.ARENA = ctx.NODE_REDUCE = None
        # ---- gradients: replicated-compute parameters count once, node-local ones stay local, the rest is summed ----
        W = self.group.world
        if W > 1:
            if self.repl_grads:
                torch._foreach_mul_(self.repl_grads, 1.0 / W)
            torch._foreach_copy_(self.keep_views, self.keep_grads)
        self.group.all_reduce_(self.gbuf)                          # [flat gradient | loss statistics]
        if W > 1:
            torch._foreach_copy_(self.keep_grads, self.keep_views)
            # global gradient norm: the optimiser kernel sees the shared part + OWN node-local part; add the other ranks' local parts
            # (reconstruction-path gradients are still unnormalised sums: scaled by 1 / kept cells like the optimiser does)
            sa = 1.0 / torch.clamp(self.stats[1], min=1.0)
            own = (self.keep_flat[:self.keep_nA] * sa).pow(2).sum().view(1)
            if phase == 1 and self.keep_nA < self.keep_flat.numel():
                own = own + self.keep_flat[self.keep_nA:].pow(2).sum()
            tot = own.clone()
            self.group.all_reduce_(tot)
            self.stats[3] += (tot - own)[0]
        self._optim()

    def _budgets(self, ada, rnd, epoch):
        return self.model.adaptive_counts(self.B * self.T * self.Ng, epoch)      # mask budgets over the GLOBAL cell count
