"""Thin Python wrappers over the C ABI (include/gptst_hip.h): one function per kernel entry point.

All tensors are contiguous fp32 CUDA tensors; calls enqueue on torch's current stream and never synchronise,
so they are hipGraph-capturable.  No CPU path exists here by design.
"""
import torch

from . import _C

MODE_TIME, MODE_NODE, MODE_SHARED = 0, 1, 2
PRO_NONE, PRO_DPRE = 0, 1
EPI_PLAIN, EPI_RES_LRELU, EPI_ADD_DPRE, EPI_LRELU, EPI_ADD_PREMUL, EPI_PREMUL = 0, 1, 2, 3, 4, 5

_p = _C.ptr


TIMER = None      # bench.py sets this to a list to collect (name, tag, start_event, end_event, bytes) per launch
LAST_CALL = {}    # (name, tag) -> argument tuple of the most recent timed launch (bench.py replays the dominant kernel)


CALL_LOCK = None  # tests that emulate ranks with threads set a threading.Lock: one C-ABI call (= all launches of one op) at a time


def _call(name, *args, tag="", nbytes=0):
    if CALL_LOCK is not None:
        with CALL_LOCK:
            _C.lib().call(name, *args, _C.stream())
        return
    if TIMER is None:
        _C.lib().call(name, *args, _C.stream())
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _C.lib().call(name, *args, _C.stream())
    e1.record()
    TIMER.append((name, tag, e0, e1, nbytes))
    LAST_CALL[(name, tag)] = args


def _nb(*ts):
    """Algorithmic bytes of a launch: every tensor operand read or written once."""
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


def _chk(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())


# ---- poolgen -------------------------------------------------------------------------------------------------
def poolgen(emb, pool, pool2=None):
    """emb (R,K); pool (K, ...) -> out (R, ...) [and out2 for pool2]."""
    _chk(emb, pool, pool2)
    R, K = emb.shape
    cols = pool.numel() // K
    out = torch.empty((R,) + tuple(pool.shape[1:]), device=emb.device, dtype=torch.float32)
    out2, cols2 = None, 0
    if pool2 is not None:
        cols2 = pool2.numel() // K
        out2 = torch.empty((R,) + tuple(pool2.shape[1:]), device=emb.device, dtype=torch.float32)
    _call("gptst_poolgen_fwd", _p(emb), _p(pool), _p(out), cols, _p(pool2), _p(out2), cols2, R, K, nbytes=_nb(emb, pool, out, pool2, out2))
    return (out, out2) if pool2 is not None else out


def poolgen_bwd_pool(emb, dW, dpool, dW2=None, dpool2=None, nsplit=1):
    """dpool += emb^T dW (summing wgrad splits); dpool2 += emb^T dW2."""
    _chk(emb, dW, dpool, dW2, dpool2)
    R, K = emb.shape
    cols = dpool.numel() // K
    cols2 = dpool2.numel() // K if dpool2 is not None else 0
    _call("gptst_poolgen_bwd_pool", _p(emb), _p(dW), _p(dpool), cols, _p(dW2), _p(dpool2), cols2, R, nsplit, K,
          nbytes=_nb(emb, dW, dpool, dW2, dpool2))


def poolgen_bwd_emb(dW, pool, demb, dW2=None, pool2=None, nsplit=1):
    """demb += dW pool^T (+ dW2 pool2^T)."""
    _chk(dW, pool, demb, dW2, pool2)
    R, K = demb.shape
    cols = pool.numel() // K
    cols2 = pool2.numel() // K if pool2 is not None else 0
    _call("gptst_poolgen_bwd_emb", _p(dW), _p(pool), cols, _p(dW2), _p(pool2), cols2, _p(demb), R, nsplit, K,
          nbytes=_nb(dW, pool, dW2, pool2, demb))


def _ptrs(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _ints(vs):
    import ctypes
    return (ctypes.c_int * len(vs))(*[int(v) for v in vs])


def poolgen_multi(emb, pools, outs=None):
    """One launch for up to 8 problems sharing emb: outs[p] (R, ...) = emb @ pools[p] (K, ...)."""
    _chk(emb, *pools)
    R, K = emb.shape
    if outs is None:
        outs = [torch.empty((R,) + tuple(pl.shape[1:]), device=emb.device, dtype=torch.float32) for pl in pools]
    cols = [pl.numel() // K for pl in pools]
    _call("gptst_poolgen_fwd_multi", _p(emb), len(pools), _ptrs(pools), _ptrs(outs), _ints(cols), R, K, nbytes=_nb(emb, *pools, *outs))
    return outs


def poolgen_bwd_pool_multi(emb, dWs, dpools, nsplits=None):
    """dpools[p] += emb^T dWs[p] (summing nsplits[p] row blocks of R rows)."""
    _chk(emb, *dWs, *dpools)
    R, K = emb.shape
    cols = [dp.numel() // K for dp in dpools]
    ns = nsplits if nsplits is not None else [1] * len(dWs)
    _call("gptst_poolgen_bwd_pool_multi", _p(emb), len(dWs), _ptrs(dWs), _ptrs(dpools), _ints(cols), _ints(ns), R, K,
          nbytes=_nb(emb, *dWs, *dpools))


def poolgen_bwd_emb_multi(dWs, pools, demb, nsplits=None):
    """demb += sum_p dWs[p] pools[p]^T."""
    _chk(demb, *dWs, *pools)
    R, K = demb.shape
    cols = [pl.numel() // K for pl in pools]
    ns = nsplits if nsplits is not None else [1] * len(dWs)
    _call("gptst_poolgen_bwd_emb_multi", len(dWs), _ptrs(dWs), _ptrs(pools), _ints(cols), _ints(ns), _p(demb), R, K,
          nbytes=_nb(demb, *dWs, *pools))


class PoolJobs:
    """A list of independent poolgen problems (forward generation and/or gradient reductions, each with its own embedding) that
    run as ONE launch per 112 jobs (gptst_pool_jobs).  Tensors are kept referenced until launch()."""
    FWD, BWD_POOL, BWD_EMB, GRAM = 0, 1, 2, 3

    def __init__(self):
        self.jobs = []
        self.post = []           # (A, G, rows, Hm): temporal graphs built by gptst_gram_fwd behind the launch (shapes beyond the job kernel)

    def fwd(self, emb, pool, out=None):
        """out (R, ...) = emb (R,K) @ pool (K, ...)"""
        _chk(emb, pool, out)
        R, K = emb.shape
        if out is None:
            out = torch.empty((R,) + tuple(pool.shape[1:]), device=emb.device, dtype=torch.float32)
        self.jobs.append((self.FWD, emb, None, pool, out, R, K, pool.numel() // K, 1, 0))
        return out

    def gram(self, emb, pool, out, A=None):
        """out (R,12,12) = A_r^T A_r, A_r = (emb @ pool)[r] viewed (cols/12, 12) — hyperTem's temporal graph (= gram_fwd of the fwd() job's output).
        A: that forward job's output (R, cols) — used when the shape does not fit the job kernel's LDS scratch (more than 20 hyperedges at
        embed_dim 16): the graph is then built by gptst_gram_fwd right behind the job launch."""
        _chk(emb, pool, out, A)
        R, K = emb.shape
        cols = pool.numel() // K
        assert cols % 12 == 0 and out.numel() == R * 144 and out.is_contiguous()
        if _C.lib().value("gptst_pool_jobs_gram_rows", K, cols) <= 0:
            assert A is not None and A.numel() == R * cols, "gram job does not fit and no forward output was given"
            self.post.append((A, out, R, cols // 12))
            return out
        self.jobs.append((self.GRAM, emb, None, pool, out, R, K, cols, 1, 0))
        return out

    @staticmethod
    def _ldx(dW, cols):
        """dW is a (rows, cols) matrix or a column window of a wider row-major one: -> its row stride"""
        assert dW.is_cuda and dW.dtype == torch.float32 and dW.dim() == 2 and dW.shape[1] == cols and dW.stride(1) == 1, (dW.shape, dW.stride(), cols)
        return dW.stride(0)

    def bwd_pool(self, emb, dW, dpool, nsplit=1):
        """dpool (K, ...) += emb^T dW, summing nsplit row blocks of R rows.  (Owned, non-atomic update: one job per dpool.)"""
        _chk(emb, dpool)
        R, K = emb.shape
        cols = dpool.numel() // K
        self.jobs.append((self.BWD_POOL, emb, dW, None, dpool, R, K, cols, nsplit, self._ldx(dW, cols)))

    def bwd_emb(self, dW, pool, demb, nsplit=1):
        """demb (R,K) += (sum of nsplit row blocks of dW) @ pool^T"""
        _chk(pool, demb)
        R, K = demb.shape
        cols = pool.numel() // K
        self.jobs.append((self.BWD_EMB, None, dW, pool, demb, R, K, cols, nsplit, self._ldx(dW, cols)))

    def launch(self):
        js, self.jobs = self.jobs, []
        post, self.post = self.post, []
        if js:
            col = lambda i: [j[i] for j in js]
            _call("gptst_pool_jobs", len(js), _ints(col(0)), _ptrs0(col(1)), _ptrs0(col(2)), _ptrs0(col(3)), _ptrs0(col(4)), _ints(col(5)),
                  _ints(col(6)), _ints(col(7)), _ints(col(8)), _ints(col(9)), nbytes=_nb(*[t for j in js for t in j[1:5]]))
        for A, G, rows, Hm in post:                      # temporal graphs whose shape is beyond the job kernel
            _call("gptst_gram_fwd", _p(A), _p(G), rows, Hm)


def _ptrs0(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])


# ---- MFMA contractions ---------------------------------------------------------------------------------------
def apply(A, W, mode, BT, N, bias=None, resid=None, A2=None, transw=False, pro=PRO_NONE, epi=EPI_PLAIN, colsum=None,
          out=None, resid2=None):
    """colsum=True -> (out, cs, ns): cs (ns*G, C) holds ns row-split partials of the column sums of pro(A) per group (the bias
    gradient; consumers sum the splits: PoolJobs.bwd_pool / bwd_emb nsplit=ns).  colsum=<tensor (G,C)>: the summed partials are
    added to it (convenience for tests)."""
    _chk(A, W, bias, resid, A2, resid2)
    C = A.shape[-1]
    if out is None:
        out = torch.empty_like(A)
    cs, ns = None, 0
    if colsum is not None:
        G = BT if mode == MODE_TIME else (N if mode == MODE_NODE else 1)
        ns = _C.lib().value("gptst_apply_nsplit", mode, BT, N, C)
        cs = torch.empty(ns * G, C, device=A.device, dtype=torch.float32)
    _call("gptst_apply", _p(A), _p(A2), _p(W), int(W.dim() == 3), int(transw), _p(bias), _p(resid), _p(resid2), _p(out), _p(cs),
          mode, pro, epi, BT, N, C, tag="mode%d pro%d epi%d" % (mode, pro, epi), nbytes=_nb(A, A2, W, bias, resid, resid2, out))
    if colsum is True:
        return out, cs, ns
    if colsum is not None:
        colsum.add_(cs.view(ns, -1, C).sum(0).view_as(colsum))
    return out


def apply_wgrad(dOut, Y, S, W, mode, BT, N, premul=False):
    """Fused backward of a generated-weight layer (C = 64): -> (dS (rows,C), dW (ns*G, C*C), dbias (ns*G, C), ns).
    Y=None: dOut already is dPre;  premul: dS is returned multiplied by lrelu'(S)."""
    _chk(dOut, Y, S, W)
    C = dOut.shape[-1]
    G = BT if mode == MODE_TIME else N
    ns = _C.lib().value("gptst_apply_wgrad_nsplit", mode, BT, N)
    dS = torch.empty_like(dOut)
    dW = torch.empty(ns * G, C * C, device=dOut.device, dtype=torch.float32)
    db = torch.empty(ns * G, C, device=dOut.device, dtype=torch.float32)
    _call("gptst_apply_wgrad", _p(dOut), _p(Y), _p(S), _p(W), _p(dS), _p(dW), _p(db), int(premul), mode, BT, N, C, tag="mode%d" % mode,
          nbytes=_nb(dOut, Y, S, W, dS, dW))
    return dS, dW, db, ns


def linear_bwd(dY, X, Wp, dOut, out, premul=False):
    """Fused backward of cap's entry Linear + the layer's residual branch (C = 64): -> (dX, dWp (ns, C*C), dbp (ns, C), ns).
    out=None: dOut already is dPre;  premul: dX is returned multiplied by lrelu'(X)."""
    _chk(dY, X, Wp, dOut, out)
    rows, C = dY.shape
    ns = _C.lib().value("gptst_linear_bwd_nsplit", rows)
    dX = torch.empty_like(dY)
    dWp = torch.empty(ns, C * C, device=dY.device, dtype=torch.float32)
    dbp = torch.empty(ns, C, device=dY.device, dtype=torch.float32)
    _call("gptst_linear_bwd", _p(dY), _p(X), _p(Wp), _p(dOut), _p(out), _p(dX), _p(dWp), _p(dbp), int(premul), rows, C,
          nbytes=_nb(dY, X, dOut, out, dX, dWp))
    return dX, dWp, dbp, ns


def wgrad_nsplit(mode, BT, N, C=64):
    return _C.lib().value("gptst_wgrad_nsplit", mode, BT, N, C)


def wgrad(A, D, mode, BT, N, D2=None, pro=PRO_NONE, colsum_a=False, colsum_d=False):
    """-> (dW (nsplit*G, C, C), nsplit);  colsum_a / colsum_d: rows become [dW (C*C) | column sums of A / of pro(D) (C)]."""
    _chk(A, D, D2)
    C = A.shape[-1]
    ns = wgrad_nsplit(mode, BT, N, C)
    G = BT if mode == MODE_TIME else (N if mode == MODE_NODE else 1)
    if colsum_a or colsum_d:
        dW = torch.empty(ns * G, C * C + C, device=A.device, dtype=torch.float32)
        _call("gptst_wgrad_colsum", _p(A), _p(D), _p(D2), _p(dW), mode, pro, 1 if colsum_a else 2, BT, N, C,
              tag="mode%d pro%d cs" % (mode, pro), nbytes=_nb(A, D, D2, dW))
    else:
        dW = torch.empty(ns * G, C, C, device=A.device, dtype=torch.float32)
        _call("gptst_wgrad", _p(A), _p(D), _p(D2), _p(dW), mode, pro, BT, N, C, tag="mode%d pro%d" % (mode, pro), nbytes=_nb(A, D, D2, dW))
    return dW, ns


# ---- temporal hypergraph -------------------------------------------------------------------------------------
def gram_fwd(A):
    N, Hm, T = A.shape
    G = torch.empty(N, T, T, device=A.device, dtype=torch.float32)
    _call("gptst_gram_fwd", _p(A), _p(G), N, Hm)
    return G


def gram_bwd(A, dG, out=None, layers=1, nsplit=1):
    """A (L*N,Hm,T); dG (L, nsplit, N, T, T) partial graph gradients (summed in a fixed order) -> dA (L*N,Hm,T)."""
    LN, Hm, T = A.shape
    dA = out if out is not None else torch.empty_like(A)
    _call("gptst_gram_bwd", _p(A), _p(dG), _p(dA), layers, LN // layers, Hm, nsplit)
    return dA


def tmix(X, G, dOut=None, Y=None):
    _chk(X, G, dOut, Y)
    B, T, N, C = X.shape
    out = torch.empty_like(X)
    _call("gptst_tmix", _p(X), _p(G), _p(dOut), _p(Y), _p(out), B, T, N, C, tag="bwd" if dOut is not None else "fwd",
          nbytes=_nb(X, G, dOut, Y, out))
    return out


def hypertem_fwd(X, G, Wbt, bbt, want_R=True):
    """Fused hyperTem forward -> (R, out), both (B,T,N,C); want_R=False -> (None, out): the backward rebuilds R from X."""
    _chk(X, G, Wbt, bbt)
    B, T, N, C = X.shape
    R = torch.empty_like(X) if want_R else None
    out = torch.empty_like(X)
    _call("gptst_hypertem_fwd", _p(X), _p(G), _p(Wbt), _p(bbt), _p(R), _p(out), B, T, N, C, nbytes=_nb(X, G, Wbt, bbt, R, out))
    return R, out


def hypertem_chain_fwd(X, stages, want_R=True):
    """Consecutive hyperTem layers in ONE launch on the (sample, 16-node) slab (gptst_hypertem_chain_fwd, C = 64).  X (B,T,N,C): input of the
    first layer;  stages: [(G, Wbt, bbt), ...] (1..3)  ->  [(R or None, out), ...], all (B,T,N,C)."""
    B, T, N, C = X.shape
    _chk(X, *[t for st in stages for t in st])
    f = lambda: torch.empty(B, T, N, C, device=X.device, dtype=torch.float32)
    Rs = [f() if want_R else None for _ in stages]
    outs = [f() for _ in stages]
    _call("gptst_hypertem_chain_fwd", _p(X), len(stages), _ptrs0([st[0] for st in stages]), _ptrs0([st[1] for st in stages]),
          _ptrs0([st[2] for st in stages]), _ptrs0(Rs), _ptrs0(outs), B, T, N, C, tag="x%d" % len(stages), nbytes=_nb(X, *Rs, *outs))
    return list(zip(Rs, outs))


def encin_ht1_fwd(source, base, mask, fill, w, bi, G, Wbt, bbt):
    """Encoder input projection + hyperTem1 on the rank-2 structure of the input (base = 1) -> out (B,T,N,C), (ab, wv) for the backward."""
    B, T, N, F = source.shape
    C = Wbt.shape[-1]
    assert base == 1
    _chk(source, mask, w, bi, G, Wbt, bbt)
    f = dict(device=source.device, dtype=torch.float32)
    out, ab, wv = torch.empty(B, T, N, C, **f), torch.empty(B * T * N, 2, **f), torch.empty(B * T, 2 * C, **f)
    _call("gptst_encin_ht1_fwd", _p(source), F, _p(mask), float(fill), _p(w), _p(bi), _p(G), _p(Wbt), _p(bbt), _p(out), _p(ab), _p(wv), B, T, N, C,
          nbytes=_nb(Wbt, out))
    return out, ab, wv


def encin_ht1_bwd(dPre, source, mask, fill, w, bi, Wbt, ab, wv, dG=None):
    """-> dWb (B*T, C*C + C) rows [dW_bt | db_bt], dG (B,N,T,T) per-sample partials, dinp (B*T, 2C) partials of d(dim_in_flow.weight | bias)."""
    B, T, N, F = source.shape
    C = Wbt.shape[-1]
    _chk(dPre, source, mask, w, bi, Wbt, ab, wv, dG)
    f = dict(device=source.device, dtype=torch.float32)
    dWb, dinp = torch.empty(B * T, C * C + C, **f), torch.empty(B * T, 2 * C, **f)
    if dG is None:
        dG = torch.empty(B, N, T, T, **f)
    _call("gptst_encin_ht1_bwd", _p(dPre), _p(source), F, _p(mask), float(fill), _p(w), _p(bi), _p(Wbt), _p(ab), _p(wv), _p(dWb), _p(dG), _p(dinp),
          B, T, N, C, nbytes=_nb(dPre, Wbt, dWb, dG))
    return dWb, dG, dinp


def guide_in_fwd(source, w1, b1, Wn, bn):
    """MLP_RL input projection + node-conditioned layer on the low-rank input form (base = 1) -> h1 (B*T*N, C)."""
    B, T, N, F = source.shape
    C = Wn.shape[-1]
    _chk(source, w1, b1, Wn, bn)
    h1 = torch.empty(B * T * N, C, device=source.device, dtype=torch.float32)
    _call("gptst_guide_in_fwd", _p(source), F, _p(w1), _p(b1), _p(Wn), _p(bn), _p(h1), B * T, N, C, nbytes=_nb(Wn, h1))
    return h1


def guide_head_fwd(source, w1, b1, Wn, bn, Wbt, bbt, W3, b3):
    """MLP_RL's whole forward on the low-rank input form (base = 1, C = 64) in two launches -> h1, h2 (B*T*N, C), prob (B*T*N, J), label int32 —
    or None when the shape needs the three launches (guide_in_fwd, apply, rowdot)."""
    B, T, N, F = source.shape
    C, J = Wn.shape[-1], W3.shape[0]
    if C != 64 or J > 16:
        return None
    _chk(source, w1, b1, Wn, bn, Wbt, bbt, W3, b3)
    dev = source.device
    f = dict(device=dev, dtype=torch.float32)
    uc = torch.empty(N, 2 * C, **f)
    h1, h2 = torch.empty(B * T * N, C, **f), torch.empty(B * T * N, C, **f)
    prob = torch.empty(B * T * N, J, **f)
    label = torch.empty(B * T * N, device=dev, dtype=torch.int32)
    _call("gptst_guide_head_fwd", _p(source), F, _p(w1), _p(b1), _p(Wn), _p(bn), _p(Wbt), _p(bbt), _p(W3), _p(b3), _p(uc), _p(h1), _p(h2), _p(prob),
          _p(label), B * T, N, C, J, nbytes=_nb(Wbt, h1, h2, prob))
    return h1, h2, prob, label


def guide_in_bwd(dPre, source, w1, b1, Wn):
    """-> dWb (N, C*C + C) rows [dW_n | db_n], dinp (N, 2C) partials of d(ln1.weight | ln1.bias)."""
    B, T, N, F = source.shape
    C = Wn.shape[-1]
    _chk(dPre, source, w1, b1, Wn)
    f = dict(device=source.device, dtype=torch.float32)
    dWb, dinp = torch.empty(N, C * C + C, **f), torch.empty(N, 2 * C, **f)
    _call("gptst_guide_in_bwd", _p(dPre), _p(source), F, _p(w1), _p(b1), _p(Wn), _p(dWb), _p(dinp), B * T, N, C, nbytes=_nb(dPre, Wn, dWb))
    return dWb, dinp


def hypertem_ntiles(N):
    return _C.lib().value("gptst_hypertem_ntiles", N)


def hypertem_bwd(dOut, Y, X, G, Wbt, dG=None, want_dbias=True, premul=False):
    """Fused hyperTem backward -> (dX, dbias (ntiles*BT, C) node-tile partials or None, dG (B,N,T,T) per-sample partials); dG may
    be a preallocated (B,N,T,T) buffer.  No atomics: consumers sum the partials (nsplit = ntiles / B).
    Y=None: dOut already is dPre = dOut*lrelu'(out);  premul: dX is returned multiplied by lrelu'(X) (the dPre of the layer below)."""
    _chk(dOut, Y, X, G, Wbt, dG)
    B, T, N, C = X.shape
    dX = torch.empty_like(X)
    nt = hypertem_ntiles(N)
    dbias = torch.empty(nt * B * T, C, device=X.device, dtype=torch.float32) if want_dbias else None
    if dG is None:
        dG = torch.empty(B, N, T, T, device=X.device, dtype=torch.float32)
    _call("gptst_hypertem_bwd", _p(dOut), _p(Y), _p(X), _p(G), _p(Wbt), _p(dX), _p(dbias), _p(dG), int(premul), B, T, N, C,
          nbytes=_nb(dOut, Y, X, G, Wbt, dX))
    return dX, dbias, dG


def hypertem_bwd_wgrad(dOut, Y, X, G, Wbt, R, dG=None, premul=False):
    """Whole backward of a hyperTem layer in ONE launch (C = 64): -> (dX, dWb (ns*BT, C*C+C) rows [dW_bt | db_bt], ns, dG (B,N,T,T)
    per-sample partials) — hypertem_bwd(want_dbias=False) and wgrad(R, dOut, MODE_TIME, D2=Y, pro=PRO_DPRE, colsum_d=True) side by side.
    Y=None / premul: as hypertem_bwd.  R=None: the weight-gradient role rebuilds R from X (one row split only)."""
    _chk(dOut, Y, X, G, Wbt, R, dG)
    B, T, N, C = X.shape
    dX = torch.empty_like(X)
    if dG is None:
        dG = torch.empty(B, N, T, T, device=X.device, dtype=torch.float32)
    ns = wgrad_nsplit(MODE_TIME, B * T, N, C)
    dWb = torch.empty(ns * B * T, C * C + C, device=X.device, dtype=torch.float32)
    _call("gptst_hypertem_bwd_wgrad", _p(dOut), _p(Y), _p(X), _p(G), _p(Wbt), _p(R), _p(dX), _p(dG), _p(dWb), int(premul), B, T, N, C,
          nbytes=_nb(dOut, Y, X, G, Wbt, R, dX, dWb))
    return dX, dWb, ns, dG


def hypertem_bwd_pair(dOut1, X1, G1, Wbt1, R1, X0, G0, Wbt0, R0, dG1, dG0, cnt, dWb1=None):
    """Backward of two consecutive hyperTem layers (1 above 0; dPre chain, both input gradients times lrelu'(input)) in ONE launch ->
    (dXmid, dX0, dWb1, dWb0, ns) or None where the pair form does not serve.  cnt: B ZEROED 32-bit words;  dWb1: optional output buffer
    (ns * B*T, C*C + C) of the upper layer's [dW_bt | db_bt] rows."""
    _chk(dOut1, X1, G1, Wbt1, R1, X0, G0, Wbt0, R0, dG1, dG0, cnt, dWb1)
    B, T, N, C = X1.shape
    if C != 64:
        return None
    dXmid, dX0 = torch.empty_like(X1), torch.empty_like(X0)
    ns = wgrad_nsplit(MODE_TIME, B * T, N, C)
    dWb0 = torch.empty(ns * B * T, C * C + C, device=X1.device, dtype=torch.float32)
    if dWb1 is None:
        dWb1 = torch.empty_like(dWb0)
    assert dWb1.shape == dWb0.shape
    try:
        _call("gptst_hypertem_bwd_pair", _p(dOut1), _p(X1), _p(G1), _p(Wbt1), _p(R1), _p(X0), _p(G0), _p(Wbt0), _p(R0), _p(dXmid), _p(dX0),
              _p(dG1), _p(dG0), _p(dWb1), _p(dWb0), _p(cnt), B, T, N, C, nbytes=_nb(dOut1, X1, R1, X0, R0, dXmid, dX0, dWb0, dWb1))
    except _C.GptstError as e:
        if e.code != _C.ESHAPE:
            raise
        return None
    return dXmid, dX0, dWb1, dWb0, ns


def tmix_bwd(dR, X, G, dOut, Y, dG=None):
    """Backward of the temporal mixing in one pass: -> (dX = dOut*lrelu'(Y) + G (*) dR, dG (N,T,T) = sum_b dR X^T)."""
    _chk(dR, X, G, dOut, Y, dG)
    B, T, N, C = X.shape
    dX = torch.empty_like(X)
    if dG is None:
        dG = torch.empty(N, T, T, device=X.device, dtype=torch.float32)
    _call("gptst_tmix_bwd", _p(dR), _p(X), _p(G), _p(dOut), _p(Y), _p(dX), _p(dG), B, T, N, C, nbytes=_nb(dR, X, dOut, Y, dX))
    return dX, dG


def tmix_bwd_chain(dR, X, G, dPre, premul=False, dG=None):
    """dPre-chain form of tmix_bwd: -> (dX = (dPre + G (*) dR) [* lrelu'(X)], dG); the layer's output is not read."""
    _chk(dR, X, G, dPre, dG)
    B, T, N, C = X.shape
    dX = torch.empty_like(X)
    if dG is None:
        dG = torch.empty(N, T, T, device=X.device, dtype=torch.float32)
    _call("gptst_tmix_bwd_chain", _p(dR), _p(X), _p(G), _p(dPre), int(bool(premul)), _p(dX), _p(dG), B, T, N, C, nbytes=_nb(dR, X, dPre, dX))
    return dX, dG


def tmix_dgraph(dR, X, out=None):
    _chk(dR, X, out)
    B, T, N, C = X.shape
    dG = out if out is not None else torch.empty(N, T, T, device=X.device, dtype=torch.float32)
    _call("gptst_tmix_dgraph", _p(dR), _p(X), _p(dG), B, T, N, C, nbytes=_nb(dR, X, dG))
    return dG


# ---- cap -----------------------------------------------------------------------------------------------------
FORCE_CAP_BIG = False      # tests: take the streaming (capbig) path even when the (b,t) capsule matrix would fit LDS


def cap_fits_lds(N, C, HS):
    return (not FORCE_CAP_BIG) and bool(_C.lib().value("gptst_cap_fits_lds", N, C, HS))


def _lds_or_stream(lds_call, stream_call):
    """Try the one-workgroup-per-(b,t) LDS kernel; when it reports that the shape does not fit (ESHAPE, nothing was launched),
    take the streaming capbig path.  (gptst_cap_fits_lds is a conservative bound over all four cap kernels together.)"""
    if FORCE_CAP_BIG:
        return stream_call()
    try:
        return lds_call()
    except _C.GptstError as e:
        if e.code != _C.ESHAPE:
            raise
        return stream_call()


def _capbig_linear(X, Wp, bp):
    """Y = X Wp^T + bp over all rows (ln_p, GPTST.py:102) on the shared-weight apply kernel."""
    B, T, N, C = X.shape
    return apply(X.reshape(-1, C), Wp, MODE_SHARED, B * T, N, bias=bp, transw=True)


def _capbig_type1(cs, P, S, BT, HS, N, C, reduce_nodes=None):
    _call("gptst_capbig_type1", _p(cs), _p(P), _p(S), BT, HS, N, C)
    if reduce_nodes is not None:            # node-sharded run: the one sum over nodes is completed across ranks here
        reduce_nodes(S)


def _cap_route_fwd_big(X, Wp, bp, dadj, HS, R, reduce_nodes=None):
    B, T, N, C = X.shape
    BT, dev = B * T, X.device
    P = _capbig_linear(X, Wp, bp)
    _call("gptst_capbig_squash_rows", _p(P), BT * N, C)
    f = dict(device=dev, dtype=torch.float32)
    cs, c = torch.empty(BT, HS, N, **f), torch.empty(BT, HS, N, **f)
    S, V0, V, s = (torch.empty(BT, HS, C, **f) for _ in range(4))
    bl = torch.zeros(BT, HS, N, **f)
    _call("gptst_capbig_softmax", None, _p(dadj), _p(cs), BT, HS, N, 0, 1)                  # c0 = softmax_h(dadj)        :105
    _capbig_type1(cs, P, S, BT, HS, N, C, reduce_nodes)
    _call("gptst_capbig_post", _p(S), None, _p(V0), BT * HS, C, 1)                          # v0 = squash(c0 . P)         :105-106
    for r in range(R):                                                                      # routing (no grad)           :113-118
        if r > 0:
            _call("gptst_capbig_type2", _p(V), _p(P), _p(bl), BT, HS, N, C)                 # b += v . P^T
        _call("gptst_capbig_softmax", _p(bl), None, _p(cs), BT, HS, N, 1, 0)
        _capbig_type1(cs, P, S, BT, HS, N, C, reduce_nodes)
        _call("gptst_capbig_post", _p(S), _p(V0), _p(V), BT * HS, C, 2)                     # v = squash(v0 (.) c.P)
    if R > 0:
        _call("gptst_capbig_type2", _p(V), _p(P), _p(bl), BT, HS, N, C)
    _call("gptst_capbig_softmax", _p(bl), _p(dadj), _p(c), BT, HS, N, 1, 1)                 # c = softmax_h(b + dadj)     :120
    _capbig_type1(c, P, S, BT, HS, N, C, reduce_nodes)
    _call("gptst_capbig_post", _p(S), None, _p(s), BT * HS, C, 0)                           # s = c . P                   :123
    return c, s


CAP_FLOW = True            # streaming cap on the fused MFMA passes of capflow.hip (False: first-generation cap_big.hip kernels)


def capflow_ok(HS, C):
    return CAP_FLOW and bool(_C.lib().value("gptst_capflow_supported", HS, C))


def _capflow_post(part, nparts, prow, V0, Vout, mode, BT, HS, C, reduce_nodes):
    """Ordered fold of the node-chunk partials + the cluster-level step; a node-sharded run completes the sums across ranks in between."""
    if reduce_nodes is not None:
        full = torch.empty(BT, prow, C, device=part.device, dtype=torch.float32)
        _call("gptst_capflow_post", _p(part), nparts, prow, None, _p(full), 3, BT, HS, C)
        reduce_nodes(full)
        part, nparts = full, 1
    _call("gptst_capflow_post", _p(part), nparts, prow, _p(V0), _p(Vout), mode, BT, HS, C)


def _cap_route_fwd_flow(X, Wp, bp, dadj, HS, R, reduce_nodes=None):
    """-> c, s, Y (pre-squash capsules, kept for the backward).  One pass over the capsule matrix per routing iteration (capflow.hip)."""
    B, T, N, C = X.shape
    BT, dev = B * T, X.device
    f = dict(device=dev, dtype=torch.float32)
    Y = _capbig_linear(X, Wp, bp)
    P = torch.empty_like(Y)
    npart = _C.lib().value("gptst_capflow_nparts", N)
    part = torch.empty(BT, npart, HS + 1, C, **f)
    c = torch.empty(BT, HS, N, **f)
    V0, V, s = (torch.empty(BT, HS, C, **f) for _ in range(3))
    _call("gptst_capflow_squash", _p(Y), _p(dadj), _p(P), _p(part), BT, HS, N, C)                          # P, c0, [c0 P ; colsum P]   :102-105
    _capflow_post(part, npart, HS + 1, V0, V if R > 0 else None, 0, BT, HS, C, reduce_nodes)               # v0; v of iteration 0        :106,:113-118
    bl = None
    for r in range(1, R):                                                                                  # b += v P^T; cs; v           :113-118
        bl2 = torch.empty(BT, HS, N, **f)
        _call("gptst_capflow_route", _p(P), _p(V), _p(bl), _p(bl2), None, None, None, _p(part), BT, HS, N, C)
        bl = bl2
        _capflow_post(part, npart, HS, V0, V, 1, BT, HS, C, reduce_nodes)
    _call("gptst_capflow_route", _p(P), _p(V) if R > 0 else None, _p(bl), None, _p(dadj), None, _p(c), _p(part), BT, HS, N, C)   # :120-123
    _capflow_post(part, npart, HS, None, s, 2, BT, HS, C, reduce_nodes)
    return c, s, Y


def cap_route_fwd(X, Wp, bp, dadj, HS, R, reduce_nodes=None, want_Y=False):
    """X (B,T,N,C); Wp (C,C) ln_p.weight; dadj (BT, HS*N) logits = teb . adj -> c (BT,HS,N), s (BT,HS,C)
    [, Y (BT*N,C) = X Wp^T + bp when the streaming path computed it, else None]."""
    _chk(X, Wp, bp, dadj)
    B, T, N, C = X.shape

    def stream():
        if capflow_ok(HS, C):
            return _cap_route_fwd_flow(X, Wp, bp, dadj, HS, R, reduce_nodes)
        return _cap_route_fwd_big(X, Wp, bp, dadj, HS, R, reduce_nodes) + (None,)

    def lds():
        c = torch.empty(B * T, HS, N, device=X.device, dtype=torch.float32)
        s = torch.empty(B * T, HS, C, device=X.device, dtype=torch.float32)
        _call("gptst_cap_route_fwd", _p(X), _p(Wp), _p(bp), _p(dadj), _p(c), _p(s), B * T, N, C, HS, R, nbytes=_nb(X, Wp, bp, dadj, c, s))
        return c, s, None
    out = stream() if reduce_nodes is not None else _lds_or_stream(lds, stream)
    return out if want_Y else out[:2]


def cap_cross_fwd(s, dyn, tmpl, B, T, HS, HT):
    _chk(s, dyn, tmpl)
    C = s.shape[-1]
    v = torch.empty_like(s)
    Ht = torch.empty(B, HT, C, device=s.device, dtype=torch.float32)
    Rt = torch.empty_like(s)
    _call("gptst_cap_cross_fwd", _p(s), _p(dyn), _p(tmpl), _p(v), _p(Ht), _p(Rt), B, T, C, HS, HT)
    return v, Ht, Rt


def cap_cross_rec_fwd(s, dyn, tmpl, c, B, T, N, HS, HT):
    """Cross-time block + cluster -> node scatter in one launch -> (v, Ht, Rt, rec) or None when the shape needs the two-launch path."""
    _chk(s, dyn, tmpl, c)
    C = s.shape[-1]
    if C != 64 or FORCE_CAP_BIG:
        return None
    v, Rt = torch.empty_like(s), torch.empty_like(s)
    Ht = torch.empty(B, HT, C, device=s.device, dtype=torch.float32)
    rec = torch.empty(B * T * N, C, device=s.device, dtype=torch.float32)
    try:
        _call("gptst_cap_cross_rec_fwd", _p(s), _p(dyn), _p(tmpl), _p(c), _p(v), _p(Ht), _p(Rt), _p(rec), B, T, N, C, HS, HT,
              nbytes=_nb(s, dyn, c, v, Rt, rec))
    except _C.GptstError as e:
        if e.code != _C.ESHAPE:
            raise
        return None
    return v, Ht, Rt, rec


def cap_cross_bwd(dv, s, Rt, Ht, dyn, tmpl, B, T, HS, HT):
    _chk(dv, s, Rt, Ht, dyn, tmpl)
    C = s.shape[-1]
    dS = torch.empty_like(s)
    ddyn = torch.empty_like(dyn)
    nws = _C.lib().value("gptst_cap_cross_ws_floats", B, T, C, HS, HT)      # 0 while the cluster tokens fit LDS
    ws = torch.empty(nws, device=s.device, dtype=torch.float32) if nws else None
    _call("gptst_cap_cross_bwd", _p(dv), _p(s), _p(Rt), _p(Ht), _p(dyn), _p(tmpl), _p(dS), _p(ddyn), _p(ws), B, T, C, HS, HT)
    return dS, ddyn


def cap_rec_fwd(c, v, N, C):
    _chk(c, v)
    BT, HS = c.shape[0], c.shape[1]
    rec = torch.empty(BT * N, C, device=c.device, dtype=torch.float32)

    def stream():
        if capflow_ok(HS, C):
            return _call("gptst_capflow_rec_fwd", _p(c), _p(v), _p(rec), BT, HS, N, C)
        return _call("gptst_capbig_rec_fwd", _p(c), _p(v), _p(rec), BT, HS, N, C)
    _lds_or_stream(lambda: _call("gptst_cap_rec_fwd", _p(c), _p(v), _p(rec), BT, N, C, HS, nbytes=_nb(c, v, rec)), stream)
    return rec


def cap_rec_bwd(drec, c, v, reduce_nodes=None):
    _chk(drec, c, v)
    BT, HS, N = c.shape
    C = v.shape[-1]
    dc1 = torch.empty_like(c)
    dv = torch.empty_like(v)

    def stream():
        if capflow_ok(HS, C):            # dc1 = drec . v^T and the partials of dv = sum_n c drec in ONE pass over drec
            npart = _C.lib().value("gptst_capflow_nparts", N)
            part = torch.empty(BT, npart, HS, C, device=c.device, dtype=torch.float32)
            _call("gptst_capflow_route", _p(drec), _p(v), None, _p(dc1), None, _p(c), None, _p(part), BT, HS, N, C)
            _capflow_post(part, npart, HS, None, dv, 2, BT, HS, C, reduce_nodes)
            return
        _call("gptst_capbig_rec_bwd_dc", _p(drec), _p(v), _p(dc1), BT, HS, N, C)
        _capbig_type1(c, drec, dv, BT, HS, N, C, reduce_nodes)                       # dv = sum_n c drec: a sum over nodes
    if reduce_nodes is not None or (HS > 16 and capflow_ok(HS, C)):     # HS > 16: the LDS kernel is the VALU first generation (172 vs 64 us at HS = 40)
        stream()
    else:
        _lds_or_stream(lambda: _call("gptst_cap_rec_bwd", _p(drec), _p(c), _p(v), _p(dc1), _p(dv), BT, N, C, HS, nbytes=_nb(drec, c, v, dc1, dv)),
                       stream)
    return dc1, dv


def cap_route_bwd(X, Wp, bp, c, dc1, dS, Y=None):
    """Y: the pre-squash capsules X Wp^T + bp when the forward kept them (streaming path); recomputed otherwise."""
    _chk(X, Wp, bp, c, dc1, dS)
    B, T, N, C = X.shape
    HS = c.shape[1]
    dY = torch.empty(B * T * N, C, device=X.device, dtype=torch.float32)
    dlogit = torch.empty_like(c)

    def stream():
        Yv = Y if Y is not None else _capbig_linear(X, Wp, bp)
        name = "gptst_capflow_route_bwd" if capflow_ok(HS, C) else "gptst_capbig_route_bwd_rows"
        _call(name, _p(Yv), _p(c), _p(dc1), _p(dS), _p(dY), _p(dlogit), B * T, HS, N, C)
    if Y is not None:
        stream()
    else:
        _lds_or_stream(lambda: _call("gptst_cap_route_bwd", _p(X), _p(Wp), _p(bp), _p(c), _p(dc1), _p(dS), _p(dY), _p(dlogit), B * T, N, C, HS,
                                     nbytes=_nb(X, Wp, bp, c, dc1, dS, dY, dlogit)), stream)
    return dY, dlogit


def cap_cross_route_bwd(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, B, T, HS, HT, flags=None):
    """Cross-time backward + routing backward in one launch -> (dY, dlogit, ddyn) or None when the shape needs the two-launch path.
    flags: 4 B ZEROED 32-bit words (a float32 tensor of zeros will do) — the cross-time backward then runs as a role of the launch (once per
    sample, overlapped with the routing workgroups' capsule GEMM) instead of a prologue repeated by every (b,t) workgroup."""
    _chk(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl)
    N, C = X.shape[2], X.shape[3]
    if C != 64 or FORCE_CAP_BIG:
        return None
    dY = torch.empty(B * T * N, C, device=X.device, dtype=torch.float32)
    dlogit = torch.empty_like(c)
    ddyn = torch.empty_like(dyn)
    try:
        dS_ws = torch.empty(B * T, HS, C, device=X.device, dtype=torch.float32) if flags is not None else None
        _call("gptst_cap_cross_route_bwd", _p(X), _p(Wp), _p(bp), _p(c), _p(dc1), _p(dv), _p(s), _p(Rt), _p(Ht), _p(dyn), _p(tmpl), _p(dY),
              _p(dlogit), _p(ddyn), _p(dS_ws), _p(flags), B, T, N, C, HS, HT, nbytes=_nb(X, Wp, bp, c, dc1, dv, s, Rt, dY, dlogit))
    except _C.GptstError as e:
        if e.code != _C.ESHAPE:
            raise
        return None
    return dY, dlogit, ddyn


def cap_cross_route_lin_bwd(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, dPre, out, premul, B, T, HS, HT, flags=None, jobs=None, nsplit=None):
    """cap_cross_route_bwd + linear_bwd in one launch (r05) -> (dX (B*T*N, C), dWp (rows, C*C), dbp (rows, C), dlogit, ddyn), or None when the shape
    needs the separate launches.  dPre: the cap layer's output gradient (already dPre when out is None; premul: dX times lrelu'(X)).
    rows = B*T + nsplit partial rows (r06: the last nsplit (b,t) run as two node halves with a partial row each; None = what the device's kernels want,
    gptst_cap_split_units).
    jobs: a PoolJobs table of gradient reductions (bwd_pool / bwd_emb) whose inputs are complete — launched here, as role workgroups of this launch where
    that form serves (gptst_cap_cross_route_lin_bwd_jobs); left untouched when None is returned."""
    _chk(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, dPre, out)
    N, C = X.shape[2], X.shape[3]
    if C != 64 or FORCE_CAP_BIG:
        return None
    dev = X.device
    if nsplit is None:
        nsplit = _C.lib().value("gptst_cap_split_units", B * T, N, C, HS)
    rows = B * T + nsplit
    dX = torch.empty(B * T * N, C, device=dev, dtype=torch.float32)
    dWp = torch.empty(rows, C * C, device=dev, dtype=torch.float32)
    dbp = torch.empty(rows, C, device=dev, dtype=torch.float32)
    dlogit = torch.empty_like(c)
    ddyn = torch.empty_like(dyn)
    try:
        dS_ws = torch.empty(B * T, HS, C, device=dev, dtype=torch.float32) if flags is not None else None
        js = jobs.jobs if jobs is not None and not jobs.post and all(j[0] in (PoolJobs.BWD_POOL, PoolJobs.BWD_EMB) for j in jobs.jobs) else []
        col = lambda i: [j[i] for j in js]      # noqa: E731
        _call("gptst_cap_cross_route_lin_bwd_split", _p(X), _p(Wp), _p(bp), _p(c), _p(dc1), _p(dv), _p(s), _p(Rt), _p(Ht), _p(dyn), _p(tmpl),
              _p(dPre), _p(out), int(bool(premul)), _p(dX), _p(dWp), _p(dbp), _p(dlogit), _p(ddyn), _p(dS_ws), _p(flags), B, T, N, C, HS, HT, nsplit,
              len(js), _ints(col(0)), _ptrs0(col(1)), _ptrs0(col(2)), _ptrs0(col(3)), _ptrs0(col(4)), _ints(col(5)), _ints(col(6)), _ints(col(7)),
              _ints(col(8)), _ints(col(9)), nbytes=_nb(X, Wp, bp, c, dc1, dv, s, Rt, dPre, out, dX, dWp, dlogit))
        if js:
            jobs.jobs = []
    except _C.GptstError as e:
        if e.code != _C.ESHAPE:
            raise
        return None
    return dX, dWp, dbp, dlogit, ddyn


# ---- mask generation (integer path) -----------------------------------------------------------------------------
_MASK_WS = {}


def _mask_ws(dev):
    """Persistent scratch for the radix-select histograms (stream-ordered reuse; allocated once per device)."""
    if dev not in _MASK_WS:
        _MASK_WS[dev] = torch.zeros(_C.lib().value("gptst_mask_ws_bytes") // 4, dtype=torch.int32, device=dev)
    return _MASK_WS[dev]


def mask_ws_floats():
    """size (in 4-byte words) of the scratch of mask_random / mask_adaptive: pass a ZEROED fp32 tensor of this size as `ws` to save the
    zeroing launch (PretrainStep takes it from the step's zero-initialised arena)."""
    return _C.lib().value("gptst_mask_ws_bytes") // 4


def _fused_jobs(jobs, u24):
    """-> the column lists of a PoolJobs table that may ride in the mask launch (gptst_mask_u24_fwd_jobs: forward and temporal-graph jobs), or None after launching
    it on its own"""
    if jobs is None or not (jobs.jobs or jobs.post):
        return None
    if not u24 or jobs.post or any(j[0] not in (PoolJobs.FWD, PoolJobs.GRAM) for j in jobs.jobs):
        jobs.launch()
        return None
    odd = [j for j in jobs.jobs if j[0] == PoolJobs.FWD and j[7] % 4]          # scalar-column forward jobs (HS * N = 2070 at METR_LA): a small launch of
    if odd:                                                                      # their own, so that the rest still rides in the mask launch
        rest = [j for j in jobs.jobs if not (j[0] == PoolJobs.FWD and j[7] % 4)]
        jobs.jobs = odd
        jobs.launch()
        jobs.jobs = rest
        if not rest:
            return None
    js, jobs.jobs = jobs.jobs, []
    col = lambda i: [j[i] for j in js]      # noqa: E731
    return js, (len(js), _ints(col(0)), _ptrs0(col(1)), _ptrs0(col(3)), _ptrs0(col(4)), _ints(col(5)), _ints(col(6)), _ints(col(7)))


def mask_random(noise, k, ws=None, u24=False, jobs=None):
    """u24: every noise value is k * 2^-24 (Philox / torch.rand) — the select runs on the integers with two digit passes instead of three
    (checked on the device: NaN mask otherwise).  jobs: a PoolJobs table of forward jobs, independent of the mask, launched here — inside the mask's
    launch when that is the cooperative one (r05)."""
    _chk(noise)
    mask = torch.empty_like(noise)
    fj = _fused_jobs(jobs, u24)
    if fj is not None:
        _call("gptst_mask_u24_fwd_jobs", 0, None, None, None, None, _p(noise), None, 0, noise.numel(), 0, 1, int(k), None, None, _p(mask),
              _p(ws if ws is not None else _mask_ws(noise.device)), int(ws is not None), *fj[1])
        return mask
    _call("gptst_mask_random_u24" if u24 else "gptst_mask_random", _p(noise), noise.numel(), int(k), _p(mask),
          _p(ws if ws is not None else _mask_ws(noise.device)), int(ws is not None))
    return mask


MASK_SMALL = 1 << 13      # cells up to which the whole mask generation is one single-workgroup launch (masksel.hip MSS_MAXM)


def labels_and_counts(prob, label):
    """(label, counts) for mask_adaptive: the guide's rowdot already produced the argmax labels and gptst_mask_adaptive histograms
    them itself (counts = None)."""
    return label, None


def mask_labels(prob):
    """prob (rows, HS) -> label int32 (rows), counts int32 (HS)."""
    _chk(prob)
    rows, HS = prob.shape
    label = torch.empty(rows, device=prob.device, dtype=torch.int32)
    counts = torch.empty(HS, device=prob.device, dtype=torch.int32)
    _call("gptst_mask_labels", _p(prob), rows, HS, _p(label), _p(counts))
    return label, counts


def mask_adaptive(label, counts, list_c, nums, noise_a, noise_r, ada_all, base, ws=None, u24=False, jobs=None):
    """-> (m_ada (M), m_rnd (M), mask (M*base)) fp32 {0,1}.  ws: see mask_ws_floats().  u24: lattice noise, jobs: see mask_random."""
    M, HS = label.numel(), list_c.numel()
    m_ada = torch.empty(M, device=label.device, dtype=torch.float32)
    m_rnd = torch.empty_like(m_ada)
    mask = torch.empty(M * base, device=label.device, dtype=torch.float32)
    fj = _fused_jobs(jobs, u24)
    if fj is not None:
        _call("gptst_mask_u24_fwd_jobs", 1, _p(label), _p(counts), _p(list_c), _p(nums), _p(noise_a), _p(noise_r), int(ada_all), M, HS, base, 0,
              _p(m_ada), _p(m_rnd), _p(mask), _p(ws if ws is not None else _mask_ws(label.device)), int(ws is not None), *fj[1])
        return m_ada, m_rnd, mask
    _call("gptst_mask_adaptive_u24" if u24 else "gptst_mask_adaptive", _p(label), _p(counts), _p(list_c), _p(nums), _p(noise_a), _p(noise_r),
          int(ada_all), M, HS, base, _p(m_ada), _p(m_rnd), _p(mask), _p(ws if ws is not None else _mask_ws(label.device)), int(ws is not None))
    return m_ada, m_rnd, mask


# ---- thin projections -------------------------------------------------------------------------------------------
def lin_in(a, lda, J, W, b, C, mask=None, fill=0.0, wlayout=0, rows=None):
    rows = rows if rows is not None else a.numel() // lda
    Y = torch.empty(rows, C, device=a.device, dtype=torch.float32)
    _call("gptst_lin_in", _p(a), lda, _p(mask), float(fill), _p(W), wlayout, _p(b), _p(Y), rows, J, C, nbytes=_nb(a, mask, W, b, Y))
    return Y


def rowdot(X, W, b, softmax=False, want_label=False):
    """Z = X W^T + b [softmax over the J outputs]; want_label -> (Z, int32 argmax per row: first maximum)."""
    rows, C = X.shape
    J = W.shape[0]
    Z = torch.empty(rows, J, device=X.device, dtype=torch.float32)
    label = torch.empty(rows, device=X.device, dtype=torch.int32) if want_label else None
    _call("gptst_rowdot", _p(X), _p(W), _p(b), _p(Z), rows, J, C, int(softmax), _p(label), nbytes=_nb(X, W, b, Z))
    return (Z, label) if want_label else Z


def rowouter(a, lda, J, X, out, olayout, csum=None, asum=None, mask=None, fill=0.0):
    rows, C = X.shape
    ws = torch.empty(_C.lib().value("gptst_rowouter_ws_floats", J, C), device=X.device, dtype=torch.float32)
    _call("gptst_rowouter", _p(a), lda, _p(mask), float(fill), _p(X), _p(out), olayout, _p(csum), _p(asum), _p(ws), rows, J, C,
          nbytes=_nb(a, mask, X, out))


def rowouter_part(a, lda, J, X, mask=None, fill=0.0, want_asum=False):
    """First stage of rowouter -> part (nparts, J*C + C + J): [sum a'^T X (j,c) | column sums of X | sums of a'] per row chunk."""
    rows, C = X.shape
    nparts = _C.lib().value("gptst_rowouter_nparts", rows)
    part = torch.empty(nparts, J * C + C + J, device=X.device, dtype=torch.float32)
    _call("gptst_rowouter_part", _p(a), lda, _p(mask), float(fill), _p(X), _p(part), int(want_asum), rows, J, C, nbytes=_nb(X))
    return part


# ---- time features ----------------------------------------------------------------------------------------------
def timefeat_fwd(params, tidx, rows, K):
    """params: the 10 nn.Linear tensors in module order; tidx (B,T,2) contiguous."""
    E = params[1].numel()
    out = torch.empty(rows, E, device=tidx.device, dtype=torch.float32)
    _call("gptst_timefeat_fwd", *[_p(t) for t in params], _p(tidx), _p(out), rows, K, E)
    return out


def timefeat_bwd(params, grads, tidx, dout, rows, K):
    E = params[1].numel()
    _call("gptst_timefeat_bwd", *[_p(t) for t in params], *[_p(t) for t in grads], _p(tidx), _p(dout), rows, K, E)


def timefeat_jobs_fwd(jobs, tidx):
    """jobs: [(params (10 tensors), rows, K)] -> list of outputs (rows, E); ONE launch."""
    outs = [torch.empty(rows, params[1].numel(), device=tidx.device, dtype=torch.float32) for params, rows, K in jobs]
    _call("gptst_timefeat_jobs", len(jobs), 0, _ptrs([t for params, _, _ in jobs for t in params]), None, _p(tidx), _ptrs(outs),
          _ints([r for _, r, _ in jobs]), _ints([k for _, _, k in jobs]), _ints([params[1].numel() for params, _, _ in jobs]))
    return outs


def timefeat_jobs_bwd(jobs, tidx):
    """jobs: [(params, grads, dout, rows, K)]; gradients are accumulated; ONE launch."""
    if not jobs:
        return
    _call("gptst_timefeat_jobs", len(jobs), 1, _ptrs([t for j in jobs for t in j[0]]), _ptrs([t for j in jobs for t in j[1]]), _p(tidx),
          _ptrs([j[2] for j in jobs]), _ints([j[3] for j in jobs]), _ints([j[4] for j in jobs]), _ints([j[0][1].numel() for j in jobs]))


# ---- loss / optimiser -------------------------------------------------------------------------------------------
def mae_fwd(out, src, lda, mask, sigma, mu, thresh, rows, J, stats):
    _call("gptst_mae_fwd", _p(out), _p(src), lda, _p(mask), float(sigma), float(mu), float(thresh), rows, J, _p(stats))


def mae_bwd(out, src, lda, mask, sigma, mu, thresh, rows, J, stats, normalize=True):
    dOut = torch.empty_like(out)
    _call("gptst_mae_bwd", _p(out), _p(src), lda, _p(mask), float(sigma), float(mu), float(thresh), rows, J, _p(stats), int(normalize),
          _p(dOut))
    return dOut


def kl(prob, c, N, w, stats, want_grad=True):
    rows, HS = prob.shape
    dlogit = torch.empty_like(prob) if want_grad else None
    _call("gptst_kl", _p(prob), _p(c), rows, N, HS, float(w), _p(dlogit), _p(stats))
    return dlogit


TAIL_MAXJ = 16


def set_deterministic(on):
    """Bit-reproducible steps: the reductions that end in float atomics by default (embedding gradients of the pool jobs, time-feature
    weight gradients) run as single-owner kernels with a fixed summation order.  Thread-local in the library."""
    _C.lib().call("gptst_set_deterministic", int(bool(on)))



def tail_parts(rows):
    return _C.lib().value("gptst_tail_parts", rows)


def tail_sws(rows, device):
    """zeroed per-workgroup loss-statistics scratch (nparts, 4) shared by tail_mae / tail_kl of one step -> stats_fold"""
    return torch.zeros(_C.lib().value("gptst_tail_parts", rows), 4, device=device, dtype=torch.float32)


def tail_mae(dec, W, b, src, lda, mask, sigma, mu, thresh, sws, premul=False):
    """Fused output head + masked-MAE + its backward (tails.hip) -> out (rows,J), d_dec (rows,C) [gradient of the SUM loss],
    part (nparts, J*C+J) partials of (gW, gb); the workgroups' (sum |y-p|, kept count) go to sws[:, 0:2] (see stats_fold)."""
    rows, C = dec.shape
    J = W.shape[0]
    nparts = _C.lib().value("gptst_tail_parts", rows)
    out = torch.empty(rows, J, device=dec.device, dtype=torch.float32)
    d_dec = torch.empty_like(dec)
    part = torch.empty(nparts, J * C + J, device=dec.device, dtype=torch.float32)
    _call("gptst_tail_mae", _p(dec), _p(W), _p(b), _p(src), lda, _p(mask), float(sigma), float(mu), float(thresh), _p(out), _p(d_dec),
          _p(part), _p(sws), int(premul), rows, J, C, nbytes=_nb(dec, d_dec))
    return out, d_dec, part


def tail_kl(h2, W3, prob, c, N, w, sws, premul=False):
    """Fused KL + softmax/ln3 backward (tails.hip) -> d_h2 (rows,C), part (nparts, HS*C+HS) partials of (gW3, gb3); KL sums -> sws[:, 2]."""
    rows, C = h2.shape
    HS = W3.shape[0]
    nparts = _C.lib().value("gptst_tail_parts", rows)
    d_h2 = torch.empty_like(h2)
    part = torch.empty(nparts, HS * C + HS, device=h2.device, dtype=torch.float32)
    _call("gptst_tail_kl", _p(h2), _p(W3), _p(prob), _p(c), float(w), _p(d_h2), _p(part), _p(sws), int(premul), rows, N, HS, C, nbytes=_nb(h2, d_h2))
    return d_h2, part


def stats_fold(sws, stats):
    """stats[0..2] += column sums of sws in a fixed order"""
    _call("gptst_stats_fold", _p(sws), sws.shape[0], _p(stats))


_ADAM_WS = {}


def step_begin(z0, z1, src, base, noise=None, rng=None):
    """Zero z0 (and z1, may be None), gather the time index of node 0 -> tidx (B,T,2), and (noise given) fill `noise` with uniform [0,1)
    numbers from Philox4x32-10 keyed by the int32 device words rng = [seed, step counter].  One launch."""
    B, T, N, lda = src.shape
    tidx = torch.empty(B, T, 2, device=src.device, dtype=torch.float32)
    _call("gptst_step_begin", _p(z0), z0.numel(), _p(z1), z1.numel() if z1 is not None else 0, _p(src), _p(tidx), B * T, N, lda, base,
          _p(noise), noise.numel() if noise is not None else 0, _p(rng))
    return tidx


def clip_adam(p, g, m, v, nA, nB, hyper, stats, ws=None, stats_out=None, sws=None):
    """clip_grad_norm_ + Adam over the flat buffers; stats[3] (in): extra squared-norm terms, stats[4] (out): total squared norm.
    ws: gptst_clip_adam_ws_floats() floats of scratch (default: one cached buffer per device — stream-ordered reuse).
    sws: the step's per-workgroup loss statistics (tail_sws): folded into stats[0..2] by the first launch instead of a separate stats_fold."""
    if ws is None:
        if p.device not in _ADAM_WS:
            _ADAM_WS[p.device] = torch.empty(_C.lib().value("gptst_clip_adam_ws_floats"), device=p.device, dtype=torch.float32)
        ws = _ADAM_WS[p.device]
    _call("gptst_clip_adam", _p(p), _p(g), _p(m), _p(v), int(nA), int(nB), _p(hyper), _p(stats), _p(ws), _p(stats_out), _p(sws),
          sws.shape[0] if sws is not None else 0)


# ---- evaluation metrics (Trainer.test) ---------------------------------------------------------------------------------
def metrics_new(T, N, device):
    return torch.zeros(T, 5, device=device, dtype=torch.float64), torch.zeros(T, N, 6, device=device, dtype=torch.float64)


def metrics_accum(out, src, lda, vis, sigma, mu, mae_thresh, mape_thresh, B, T, N, D, sums_t, sums_tn):
    """Add one batch to the metric sums: out (B*T*N, D), src (B,T,N,lda) normalised input (label = first D channels), vis fp32 mask or None."""
    _call("gptst_metrics_accum", _p(out), _p(src), lda, _p(vis), float(sigma), float(mu), int(mae_thresh is not None),
          float(mae_thresh if mae_thresh is not None else 0.0), float(mape_thresh), B, T, N, D, _p(sums_t), _p(sums_tn))


def metrics_report(sums_t, sums_tn):
    """-> (T+1, 4) float64 rows [mae, rmse, mape, corr]: one per horizon, then the average over all horizons
    (BasicTrainer.py:241-248; CORR per lib/metrics.py:52-77: per node over (batch, [time,] channel), unbiased std, nodes with
    constant truth skipped, mean over nodes)."""
    st, sn = sums_t.cpu(), sums_tn.cpu()

    def rows_of(a, m):
        mae, rmse, mape = a[..., 1] / a[..., 0], torch.sqrt(a[..., 2] / a[..., 0]), a[..., 4] / a[..., 3]
        K, sp, sy, spp, syy, spy = (m[..., k] for k in range(6))
        pm, ym = sp / K, sy / K
        cov = spy / K - pm * ym
        pvar, yvar = (spp - K * pm * pm) / (K - 1), (syy - K * ym * ym) / (K - 1)
        c = cov / torch.sqrt(pvar.clamp_min(0) * yvar.clamp_min(0))
        ok = yvar > 1e-9 * (syy / K).clamp_min(1e-30)                     # true_std != 0
        corr = torch.stack([c[i][ok[i]].mean() for i in range(c.shape[0])]) if c.dim() == 2 else c[ok].mean()
        return torch.stack([mae, rmse, mape, corr], -1)

    per_t = rows_of(st, sn)
    avg = rows_of(st.sum(0), sn.sum(0))
    return torch.cat([per_t, avg[None]], 0)
