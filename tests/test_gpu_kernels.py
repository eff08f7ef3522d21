"""GPU parity tests (run on the MI355X box: pytest -m gpu): HIP kernels through the C ABI vs the pinned CPU oracle.
Tolerance: fp32 1e-4 (north-star) on BOTH the max error relative to the tensor scale and the element-wise relative error (2 % floor),
for every kernel output and gradient; measured worst values are recorded (profiles/parity_r02.json); index/mask work bit-exact."""
import os

import pytest
import torch

from oracle import gptst_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


ELEM_FLOOR = 0.02      # element-wise relative error is taken against |ref| + ELEM_FLOOR * max|ref|
ELEM_OUTLIERS = {"cap dadj": 1.5e-4, "cap dt_adj": 1.5e-4}      # measured 9.1e-5 / 9.1e-5 (test_cap_layer_streaming_path, N = 1030 / HS = 64), see close()


def close(a, b, tol=1e-4, what=""):
    """fp32 parity bound (north-star: 1e-4 relative): BOTH the max abs error relative to the tensor scale and the element-wise
    relative error |a-b| / (|b| + 2 % of the scale) must stay below tol; the measured values go to gpurun_out/parity.json."""
    from conftest import record_current
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = b.abs().max().clamp_min(1e-6)
    err = float((a - b).abs().max() / scale)
    elem = float(((a - b).abs() / (b.abs() + ELEM_FLOOR * scale)).max())
    key = (what or "tensor").split(" [")[0]
    record_current("scaled:" + key, err)
    record_current("elem:" + key, elem)
    if os.environ.get("GPTST_SOFT_CLOSE") and not (err < tol):        # debugging aid: report where a tensor is off instead of stopping
        d = (a - b).abs() / scale
        bad = (d > tol).nonzero()
        print("SOFT FAIL %s: err %.3e, %d of %d elements, first %s last %s, shape %s"
              % (what, err, len(bad), d.numel(), bad[:2].tolist(), bad[-2:].tolist(), tuple(d.shape)))
        return err
    assert err < tol, "%s: max err / scale = %.3e (scale %.3e)" % (what, err, float(scale))
    # r04: the element-wise bound is the north-star 1e-4 too (r03: 2e-4).  Measured over the whole suite (profiles/parity_r04.json, 1882 tensors):
    # scaled worst 1.3e-5; element-wise worst 9.1e-5, and only the two tensors named below come within 25 % of the bound — both are gradients
    # of the cap's cluster logits / cross-time graph on the STREAMING path at shapes where a row sums 1030 nodes or 64 clusters in chunk
    # order (the oracle sums them in one torch.matmul); they keep 1.5e-4 so that a different-but-valid summation order does not flip the test.
    ebound = max(1e-4, tol) if key not in ELEM_OUTLIERS else max(ELEM_OUTLIERS[key], tol)
    assert elem < ebound, "%s: element-wise rel err = %.3e (floor %.0e x scale)" % (what, elem, ELEM_FLOOR)
    return err


def rnd(*s, g, scale=1.0):
    return torch.randn(*s, generator=g) * scale


def test_poolgen_fwd_bwd():
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    for R, K, cols, cols2 in ((384, 16, 4096, 64), (170, 16, 4096, 64), (24, 4, 2070, 0), (3, 4, 1920, 0), (20, 8, 96, 0)):
        emb, pool = rnd(R, K, g=g), rnd(K, cols, g=g)
        pool2 = rnd(K, cols2, g=g) if cols2 else None
        r = ops.poolgen(emb.to(dev), pool.to(dev), pool2.to(dev) if cols2 else None)
        out = r[0] if cols2 else r
        close(out, emb @ pool, what="poolgen out")
        if cols2:
            close(r[1], emb @ pool2, what="poolgen out2")
        for ns in (1, 3):
            dW = rnd(ns * R, cols, g=g)
            dW2 = rnd(R, cols2, g=g) if cols2 else None
            dpool = torch.zeros(K, cols, device=dev)
            dpool2 = torch.zeros(K, cols2, device=dev) if cols2 else None
            ops.poolgen_bwd_pool(emb.to(dev), dW.to(dev), dpool, dW2.to(dev) if cols2 else None, dpool2, nsplit=ns)
            dWs = dW.view(ns, R, cols).sum(0)
            close(dpool, emb.t() @ dWs, what="dpool")
            if cols2:
                close(dpool2, emb.t() @ dW2, what="dpool2")
            demb = torch.ones(R, K, device=dev)
            ops.poolgen_bwd_emb(dW.to(dev), pool.to(dev), demb, dW2.to(dev) if cols2 else None,
                                pool2.to(dev) if cols2 else None, nsplit=ns)
            ref = 1 + dWs @ pool.t() + (dW2 @ pool2.t() if cols2 else 0)
            close(demb, ref, what="demb")


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("BT,N,C", [(24, 20, 64), (36, 170, 64), (5, 33, 64), (24, 20, 128), (12, 300, 128), (5, 33, 128), (3, 1100, 128)])
def test_apply_and_wgrad(mode, BT, N, C):
    """MFMA contractions vs fp32 matmul, asymmetric random weights (catches transposed fragments).  C = 128: apply128 / wgrad128
    (second generation; BASELINE configs[4]) incl. several workgroups per group, ragged tiles and row splits."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(7 + mode)
    G = BT if mode == 0 else (N if mode == 1 else 1)
    A = rnd(BT, N, C, g=g); res = rnd(BT, N, C, g=g)
    W = rnd(G, C, C, g=g, scale=0.2); bias = rnd(G, C, g=g)
    Wd = W if mode != 2 else W[0]

    def ref_apply(A_, W_):
        if mode == 0:
            return torch.einsum("gni,gio->gno", A_, W_)
        if mode == 1:
            return torch.einsum("bni,nio->bno", A_, W_)
        return A_ @ W_[0]

    def bias_b():
        return bias.view(BT, 1, C) if mode == 0 else (bias.view(1, N, C) if mode == 1 else bias.view(1, 1, C))

    out = ops.apply(A.to(dev), Wd.to(dev).contiguous(), mode, BT, N, bias=bias.to(dev), resid=res.to(dev), epi=ops.EPI_RES_LRELU)
    ref = torch.nn.functional.leaky_relu(ref_apply(A, W) + bias_b() + res, 0.01).contiguous()
    close(out, ref, what="apply fwd")
    # plain + transposed weight
    out = ops.apply(A.to(dev), Wd.to(dev).contiguous(), mode, BT, N, transw=True)
    close(out, ref_apply(A, W.transpose(1, 2)), what="apply transw")
    # backward-data with dPre prologue and column sums
    dout = rnd(BT, N, C, g=g)
    dpre = dout * torch.where(ref > 0, 1.0, 0.01)
    cs = torch.zeros(G, C, device=dev)
    dA = ops.apply(dout.to(dev), Wd.to(dev).contiguous(), mode, BT, N, A2=ref.to(dev), transw=True, pro=ops.PRO_DPRE, colsum=cs)
    close(dA, ref_apply(dpre, W.transpose(1, 2)), what="apply bwd data")
    cs_ref = dpre.sum(1) if mode == 0 else (dpre.sum(0) if mode == 1 else dpre.sum((0, 1)).view(1, C))
    close(cs, cs_ref, what="colsum")
    # weight gradient
    dW, ns = ops.wgrad(A.to(dev), dout.to(dev), mode, BT, N, D2=ref.to(dev), pro=ops.PRO_DPRE)
    dW = dW.view(ns, G, C, C).sum(0)
    if mode == 0:
        dW_ref = torch.einsum("gni,gno->gio", A, dpre)
    elif mode == 1:
        dW_ref = torch.einsum("bni,bno->nio", A, dpre)
    else:
        dW_ref = torch.einsum("bni,bno->io", A, dpre).view(1, C, C)
    close(dW, dW_ref, what="wgrad")


@pytest.mark.parametrize("B,N,C", [(3, 21, 64), (5, 9, 128), (1, 3, 128)])
def test_tmix_and_graph(B, N, C):
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    T, Hm = 12, 8
    A = rnd(N, Hm, T, g=g); X = rnd(B, T, N, C, g=g); dR = rnd(B, T, N, C, g=g)
    G = ops.gram_fwd(A.to(dev))
    Gref = torch.einsum("nht,nhu->ntu", A, A)
    close(G, Gref, what="gram")
    out = ops.tmix(X.to(dev), G)
    close(out, torch.einsum("ntu,bunc->btnc", Gref, X), what="tmix")
    Y = rnd(B, T, N, C, g=g); dO = rnd(B, T, N, C, g=g)
    out = ops.tmix(X.to(dev), G, dOut=dO.to(dev), Y=Y.to(dev))
    close(out, torch.einsum("ntu,bunc->btnc", Gref, X) + dO * torch.where(Y > 0, 1.0, 0.01), what="tmix+dpre")
    dG = ops.tmix_dgraph(dR.to(dev), X.to(dev))
    close(dG, torch.einsum("btnc,bunc->ntu", dR, X), what="dgraph")
    # both in one pass over dR (general, non-symmetric G: dX[u] = sum_t G[t,u] dR[t])
    Gn = rnd(N, T, T, g=g)
    dX1, dG1 = ops.tmix_bwd(dR.to(dev), X.to(dev), Gn.to(dev), dO.to(dev), Y.to(dev))
    close(dX1, torch.einsum("ntu,btnc->bunc", Gn, dR) + dO * torch.where(Y > 0, 1.0, 0.01), what="tmix_bwd dX")
    close(dG1, torch.einsum("btnc,bunc->ntu", dR, X), what="tmix_bwd dG")
    dGr = rnd(N, T, T, g=g)
    dA = ops.gram_bwd(A.to(dev), dGr.to(dev))
    Ar = A.clone().requires_grad_()
    (torch.einsum("nht,nhu->ntu", Ar, Ar) * dGr).sum().backward()
    close(dA, Ar.grad, what="gram bwd")


def _hypertem_case(B, N, C, d, Hm, seed):
    g = torch.Generator().manual_seed(seed)
    x = rnd(B, 12, N, C, g=g); ne = rnd(N, d, g=g); te = rnd(B, 12, d, g=g)
    adj = rnd(d, Hm, 12, g=g, scale=0.3); wp = rnd(d, C, C, g=g, scale=0.1); bp = rnd(d, C, g=g, scale=0.3)
    go = rnd(B, 12, N, C, g=g)
    return [x, ne, te, adj, wp, bp], go


@pytest.mark.parametrize("B,N,d,Hm", [(2, 20, 8, 8), (3, 170, 16, 8), (1, 33, 4, 5)])
def test_hypertem_layer(B, N, d, Hm):
    from gptst_amd import layers
    dev = _dev()
    ts, go = _hypertem_case(B, N, 64, d, Hm, 11)
    cpu = [t.clone().requires_grad_() for t in ts]
    sd = {"h.adj": cpu[3], "h.weights_pool": cpu[4], "h.bias_pool": cpu[5]}
    ref = O.hypertem(sd, "h.", cpu[0], cpu[1], cpu[2])
    go = go * (ref.detach().abs() > 1e-5)       # a pre-activation within fp32 noise of 0 may pick the other LReLU slope: not a parity question
    (ref * go).sum().backward()
    gpu = [t.to(dev).requires_grad_() for t in ts]
    out = layers.hypertem(*gpu)
    (out * go.to(dev)).sum().backward()
    close(out, ref, what="hypertem out")
    for nm, a, b in zip(["x", "node_emb", "time_eb", "adj", "wpool", "bpool"], gpu, cpu):
        close(a.grad, b.grad, what="hypertem d" + nm)


def _cap_case(B, N, C, d, ds, HS, HT, seed):
    g = torch.Generator().manual_seed(seed)
    T = 12
    x = rnd(B, T, N, C, g=g, scale=0.5); ne = rnd(N, d, g=g); tes = rnd(B, ds, g=g); teb = rnd(B, T, ds, g=g)
    t_adj = rnd(ds, HT, T * HS, g=g, scale=0.2); adj = rnd(ds, HS, N, g=g, scale=0.5)
    wspa = rnd(d, C, C, g=g, scale=0.1); bspa = rnd(d, C, g=g, scale=0.3)
    lw = rnd(C, C, g=g, scale=0.15); lb = rnd(C, g=g, scale=0.3)
    go = rnd(B, T, N, C, g=g)
    return [x, ne, tes, teb, t_adj, adj, wspa, bspa, lw, lb], go


@pytest.mark.parametrize("B,N,d,ds,HS,HT,R", [(2, 20, 8, 4, 5, 6, 3), (2, 170, 16, 4, 10, 16, 2), (1, 33, 4, 3, 16, 5, 0),
                                              (1, 41, 4, 4, 20, 8, 2)])
@pytest.mark.parametrize("one_launch", [False, True], ids=["kernels", "route_lin_bwd"])
def test_cap_layer(B, N, d, ds, HS, HT, R, one_launch, monkeypatch):
    """one_launch (r05): the backward through gptst_cap_cross_route_lin_bwd — cross-time role + routing backward + the entry Linear's backward in one
    launch, the fused step's default — against the ORACLE (dx, ln_p weight / bias, cluster-logit and cross-time-graph gradients)."""
    from gptst_amd import layers
    monkeypatch.setattr(layers, "CAP_BWD_ONE_LAUNCH", one_launch)
    dev = _dev()
    C, T = 64, 12
    ts, go = _cap_case(B, N, C, d, ds, HS, HT, 21)
    tmpl = torch.linspace(1, T, steps=T) / 12.0
    cpu = [t.clone().requires_grad_() for t in ts]
    sd = {"c.t_adj": cpu[4], "c.adj": cpu[5], "c.weights_spa": cpu[6], "c.bias_spa": cpu[7], "c.ln_p.weight": cpu[8],
          "c.ln_p.bias": cpu[9], "c.mask_template": tmpl}
    ref, cref, dynref, aux = O.cap(sd, "c.", cpu[0], cpu[1], cpu[2], cpu[3], R, materialize_5d=(N <= 64), return_aux=True)
    go = go * (ref.detach().abs() > 1e-5)       # a pre-activation within fp32 noise of 0 may pick the other LReLU slope: not a parity question
    (ref * go).sum().backward()
    gpu = [t.to(dev).requires_grad_() for t in ts]
    out, c, dyn = layers.cap(*gpu, tmpl.to(dev), R)
    (out * go.to(dev)).sum().backward()
    close(c, cref.squeeze(-1), what="cap c")
    close(dyn, dynref, what="cap dyn")
    close(out, ref, what="cap out")
    names = ["x", "node_emb", "time_eb_spg", "teb", "t_adj", "adj", "wspa", "bspa", "lnp_w", "lnp_b"]
    for nm, a, b in zip(names, gpu, cpu):
        close(a.grad, b.grad, what="cap d" + nm)


@pytest.mark.parametrize("flow", [True, False], ids=["capflow", "capbig"])
@pytest.mark.parametrize("B,N,C,d,ds,HS,HT,R,force", [(2, 20, 64, 8, 4, 5, 6, 3, True), (1, 170, 64, 16, 4, 10, 16, 2, True),
                                                       (1, 600, 64, 8, 4, 10, 16, 2, False), (1, 300, 128, 8, 4, 10, 8, 2, False),
                                                       (1, 37, 128, 4, 3, 40, 5, 1, True), (2, 37, 64, 4, 3, 7, 5, 1, True),
                                                       (1, 530, 128, 4, 3, 16, 5, 0, True), (1, 1030, 64, 4, 3, 3, 5, 4, False),
                                                       (1, 266, 64, 4, 3, 40, 5, 2, True), (2, 70, 64, 4, 3, 20, 5, 2, True), (1, 45, 64, 4, 3, 64, 5, 1, True),
                                                       (1, 33, 64, 4, 3, 17, 5, 3, True)])
def test_cap_layer_streaming_path(B, N, C, d, ds, HS, HT, R, force, flow):
    """cap through the streaming kernels — the fused MFMA passes of capflow.hip (HS <= 16) or the first-generation cap_big.hip kernels:
    taken when the (b,t) capsule matrix does not fit LDS (N = 600 at C = 64, N = 300 at C = 128 — BASELINE config 5 territory) or
    forced, vs the oracle; the LDS path is covered by test_cap_layer.  Ragged node counts (N % 4 != 0, N % 16 != 0, several 256-node
    chunks) and R = 0 .. 4 routing iterations included."""
    from gptst_amd import layers, ops
    dev = _dev()
    T = 12
    assert force or not ops.cap_fits_lds(N, C, HS)
    ts, go = _cap_case(B, N, C, d, ds, HS, HT, 33)
    tmpl = torch.linspace(1, T, steps=T) / 12.0
    cpu = [t.clone().requires_grad_() for t in ts]
    sd = {"c.t_adj": cpu[4], "c.adj": cpu[5], "c.weights_spa": cpu[6], "c.bias_spa": cpu[7], "c.ln_p.weight": cpu[8],
          "c.ln_p.bias": cpu[9], "c.mask_template": tmpl}
    ref, cref, dynref, aux = O.cap(sd, "c.", cpu[0], cpu[1], cpu[2], cpu[3], R, materialize_5d=False, return_aux=True)
    go = go * (ref.detach().abs() > 1e-5)       # a pre-activation within fp32 noise of 0 may pick the other LReLU slope: not a parity question
    (ref * go).sum().backward()
    gpu = [t.to(dev).requires_grad_() for t in ts]
    ops.FORCE_CAP_BIG, ops.CAP_FLOW = force, flow
    try:
        out, c, dyn = layers.cap(*gpu, tmpl.to(dev), R)
        (out * go.to(dev)).sum().backward()
    finally:
        ops.FORCE_CAP_BIG, ops.CAP_FLOW = False, True
    close(c, cref.squeeze(-1), what="cap c")
    close(out, ref, what="cap out")
    names = ["x", "node_emb", "time_eb_spg", "teb", "t_adj", "adj", "wspa", "bspa", "lnp_w", "lnp_b"]
    for nm, a, b in zip(names, gpu, cpu):
        close(a.grad, b.grad, what="cap d" + nm)


# ---------------------------------------------------------------------------------------------------------------
# integer path: mask generation, bit-exact vs the oracle's sort/scatter restatement
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,ratio,seed", [(65280, 0.25, 1), (102144, 0.25, 6), (5000, 0.25, 2), (1234, 0.9, 3), (777, 0.0, 4), (300, 1.0, 5)])
def test_mask_random_bit_exact(M, ratio, seed):
    from gptst_amd import ops, synth
    dev = _dev()
    noise = synth.make_noise(M, seed)
    ref = O.random_mask(noise, ratio)
    for multi in (0, 1):                  # single-workgroup launch (M <= 2^18) and the multi-launch radix select
        with _mask_path(multi):
            got = ops.mask_random(noise.to(dev), int(M * ratio))
        assert torch.equal(got.cpu().to(torch.int64), ref), multi
    # r04: lattice noise (k * 2^-24, what synth / torch.rand / the step's Philox draw) -> the select on the integers: two digit passes (multi-
    # workgroup form; <= 8192 cells: one launch of one workgroup) and (path 2) the one-workgroup form at every size up to 65536 cells
    for path in (0, 1, 2, 3):
        with _mask_path(path):
            got = ops.mask_random(noise.to(dev), int(M * ratio), u24=True)
        assert torch.equal(got.cpu().to(torch.int64), ref), ("u24", path)


def _handoff_timeouts():
    import ctypes
    from gptst_amd import _C
    n = ctypes.c_int(0)
    _C.lib().call("gptst_handoff_timeouts", ctypes.byref(n))
    return n.value


class _mask_path:
    """mask generation path: 0 by size (lattice noise: one workgroup <= 8192 cells, the cooperative launch <= 131072, else multi-launch),
    1 multi-launch, 2 one workgroup at every size it covers, 3 as 0 without the cooperative launch"""

    def __init__(self, multi):
        self.multi = multi

    def __enter__(self):
        from gptst_amd import _C
        _C.lib().call("gptst_mask_force_multi", 0 if self.multi == 3 else int(self.multi))
        _C.lib().call("gptst_mask_cooperative", 0 if self.multi == 3 else -1)

    def __exit__(self, *a):
        from gptst_amd import _C
        _C.lib().call("gptst_mask_force_multi", 0)
        _C.lib().call("gptst_mask_cooperative", -1)


@pytest.mark.parametrize("multi", [0, 1])
@pytest.mark.parametrize("reps", [40, 4000, 12500])
def test_mask_random_ties_lowest_index(multi, reps):
    """Duplicated values straddling rank k: exactly k cells are dropped, larger values first, ties -> lowest index.  reps = 4000
    spreads the tied cells over all 64 workgroups of the multi-launch path (tied cells are written by ONE workgroup)."""
    from gptst_amd import ops
    dev = _dev()
    noise = torch.tensor([0.5, 0.25, 0.5, 0.75, 0.5, 0.1, 0.5, 0.0] * reps)
    n = noise.numel()
    order = torch.sort(-noise, stable=True)[1]                 # descending value, ties -> lowest index
    for k in (0, reps, reps + 1, reps * 5 // 2, 5 * reps - 1, 5 * reps, 5 * reps + 1, 8 * reps):
        with _mask_path(multi):
            for _ in range(3 if reps > 40 else 1):             # the old cross-workgroup race was timing dependent
                got = ops.mask_random(noise.to(dev), k).cpu()
                assert int((got == 0).sum()) == k
                ref = torch.ones(n)
                ref[order[:k]] = 0
                assert torch.equal(got, ref), k
        lat = torch.where(noise == 0.1, torch.tensor(0.125), noise)     # lattice paths: 0.1 is not k * 2^-24 -> 0.125 (same order, same ties)
        for path in ((1,) if multi else (2, 0)):               # 1: two digit passes over many workgroups; 2: the one-workgroup form; 0: by size (r05: 32000 cells -> the cooperative launch)
            with _mask_path(path):
                assert torch.equal(ops.mask_random(lat.to(dev), k, u24=True).cpu(), ref), ("u24", path, k)


@pytest.mark.parametrize("path,M", [(0, 5000), (0, 40000), (1, 40000), (2, 40000), (3, 40000)])
def test_mask_u24_rejects_noise_off_the_lattice(path, M):
    """gptst_mask_*_u24 select on the integers k = noise * 2^24: a value that is not k * 2^-24 (or outside [0,1)) must not be rounded silently —
    the whole mask comes back NaN; lattice noise gives a clean {0,1} mask.  path 0: one workgroup (<= 8192 cells), 1: two digit passes over
    many workgroups, 2: one workgroup at any size up to 65536; path 0 at 40000 cells: the cooperative launch (r05), 3: the same size without it."""
    from gptst_amd import ops, synth
    dev = _dev()
    noise = synth.make_noise(M, 21)
    assert torch.equal((noise * 2 ** 24).round() / 2 ** 24, noise) and float(noise.max()) < 1
    with _mask_path(path):
        ok = ops.mask_random(noise.to(dev), M // 4, u24=True)
        assert set(ok.unique().tolist()) == {0.0, 1.0}
        for bad in (0.1, 1.0, -0.25, float("nan")):              # 0.1 is not on the lattice; 1.0 / negative / NaN are outside [0, 1)
            nz = noise.clone(); nz[1234] = bad
            assert torch.isnan(ops.mask_random(nz.to(dev), M // 4, u24=True)).all(), bad
        lab = torch.randint(0, 10, (M,), dtype=torch.int32).to(dev)
        lc = torch.tensor(synth.class_order(10, 1), dtype=torch.int32, device=dev)
        nums = torch.tensor([M // 8, M // 8], dtype=torch.int32, device=dev)
        for which in (0, 1):
            nz = noise.clone(); nz[77] = 0.25 + 2.0 ** -25        # a float32 below 0.5 with an odd last mantissa bit: not on the 2^-24 lattice
            na, nr = (nz, noise) if which == 0 else (noise, nz)
            assert torch.isnan(ops.mask_adaptive(lab, None, lc, nums, na.to(dev), nr.to(dev), 1, 1, u24=True)[2]).all(), which
        clean = ops.mask_adaptive(lab, None, lc, nums, noise.to(dev), synth.make_noise(M, 22).to(dev), 1, 1, u24=True)[2]
        assert set(clean.unique().tolist()) == {0.0, 1.0} and int((clean == 0).sum()) == 2 * (M // 8)
    for n_big in (70000, 140000):            # 70000: the cooperative launch on 69 workgroups (path 0) / the digit passes; 140000: beyond the cooperative form
        big = synth.make_noise(n_big, 5)
        assert torch.equal(ops.mask_random(big.to(dev), n_big // 4, u24=True).cpu().long(), O.random_mask(big, 0.25)), n_big


@pytest.mark.parametrize("ada_all", [1, 0])
@pytest.mark.parametrize("B,N,HS,frac", [(32, 170, 10, 0.5), (32, 266, 10, 0.4), (4, 20, 5, 0.9), (3, 17, 10, 0.01), (2, 33, 16, 0.0)])
def test_mask_adaptive_bit_exact(ada_all, B, N, HS, frac):
    from gptst_amd import ops, synth
    dev = _dev()
    T = 12
    M = B * T * N
    g = torch.Generator().manual_seed(B * 100 + HS)
    prob = torch.softmax(torch.randn(M, HS, generator=g) * 2, -1)
    label_ref = torch.sort(prob, dim=-1, descending=True)[1][..., 0]
    label, counts = ops.mask_labels(prob.to(dev))
    assert torch.equal(label.cpu().long(), label_ref)
    assert torch.equal(counts.cpu().long(), torch.bincount(label_ref, minlength=HS))
    total = int(M * 0.25)
    ada = int(total * frac)
    rnd_n = total - ada
    list_c = synth.class_order(HS, 5)
    na, nr = synth.make_noise(M, 11), synth.make_noise(M, 12)
    m_ada_r, m_rnd_r, fin_r = O.adaptive_mask(label_ref.view(B, T, N), list_c, na, nr, ada, rnd_n, "all" if ada_all else "half")
    for base, multi, cnt in ((1, 0, counts), (2, 0, None), (1, 1, counts), (2, 1, None)):      # cnt None: class histogram taken inside
      with _mask_path(multi):
        m_ada, m_rnd, mask = ops.mask_adaptive(label, cnt, torch.tensor(list_c, dtype=torch.int32, device=dev),
                                               torch.tensor([ada, rnd_n], dtype=torch.int32, device=dev), na.to(dev), nr.to(dev),
                                               ada_all, base)
        assert torch.equal(m_ada.cpu().long(), m_ada_r)
        assert torch.equal(m_rnd.cpu().long(), m_rnd_r)
        assert torch.equal(mask.cpu().long().view(M, base), fin_r.view(M, 1).repeat(1, base))
        assert int((mask.view(M, base)[:, 0] == 0).sum()) == total
    for base, path, cnt in ((1, 0, counts), (2, 0, None), (2, 1, None), (1, 1, counts), (2, 2, None), (1, 3, counts)):   # lattice noise (0: by size - r05: the cooperative launch at M > 8192, 1: two digit passes, 2: one workgroup, 3: by size without the cooperative launch)
      with _mask_path(path):
        m_ada, m_rnd, mask = ops.mask_adaptive(label, cnt, torch.tensor(list_c, dtype=torch.int32, device=dev),
                                               torch.tensor([ada, rnd_n], dtype=torch.int32, device=dev), na.to(dev), nr.to(dev),
                                               ada_all, base, u24=True)
        assert torch.equal(m_ada.cpu().long(), m_ada_r) and torch.equal(m_rnd.cpu().long(), m_rnd_r), ("u24", path)
        assert torch.equal(mask.cpu().long().view(M, base), fin_r.view(M, 1).repeat(1, base)), ("u24", path)


@pytest.mark.parametrize("adaptive", [0, 1])
@pytest.mark.parametrize("coop", [1, 0])
def test_mask_launch_carries_forward_jobs(adaptive, coop):
    """gptst_mask_u24_fwd_jobs (r05): a table of forward generation jobs rides in the cooperative mask launch (coop = 0: the same call falls back to
    gptst_pool_jobs + the multi-launch mask).  Masks and job outputs are bit-identical to the separate calls; a job table that is not forward-only or
    not float4-shaped goes its own way."""
    from gptst_amd import ops, synth
    dev = _dev()
    B, T, N, HS, base = 32, 12, 170, 10, 1
    M = B * T * N
    g = torch.Generator().manual_seed(5)
    label = torch.randint(0, HS, (M,), generator=g).to(torch.int32).to(dev)
    list_c = torch.tensor(synth.class_order(HS, 2), dtype=torch.int32, device=dev)
    total = int(M * 0.25)
    nums = torch.tensor([total // 2, total - total // 2], dtype=torch.int32, device=dev)
    na, nr = synth.make_noise(M, 31).to(dev), synth.make_noise(M, 32).to(dev)
    shapes = [(384, 10, 4096), (384, 10, 64), (170, 10, 16 * 12), (170, 5, 4096), (32, 8, 36 * 12), (384, 10, 1700), (33, 3, 8)]
    embs = [torch.randn(R, K, generator=g).to(dev) for R, K, _ in shapes]
    pools = [torch.randn(K, c, generator=g).to(dev) for _, K, c in shapes]

    def table(extra=()):
        pj = ops.PoolJobs()
        outs = [pj.fwd(e, p_) for e, p_ in zip(embs, pools)]
        for k in (2, 4):                                   # temporal graphs (kind 3: whole workgroups of the launch) of the two cols % 12 == 0 shapes
            outs.append(pj.gram(embs[k], pools[k], torch.empty(shapes[k][0], 12, 12, device=dev)))
        for e, p_ in extra:
            outs.append(pj.fwd(e, p_))
        return pj, outs

    pj, ref_out = table()
    pj.launch()
    with _mask_path(0 if coop else 3):
        ref_mask = (ops.mask_adaptive(label, None, list_c, nums, na, nr, 1, base, u24=True) if adaptive else (ops.mask_random(na, total, u24=True),))
        for rep in range(3):
            pj, outs = table()
            got = (ops.mask_adaptive(label, None, list_c, nums, na, nr, 1, base, u24=True, jobs=pj) if adaptive
                   else (ops.mask_random(na, total, u24=True, jobs=pj),))
            assert not pj.jobs
            for a, b in zip(got, ref_mask):
                assert torch.equal(a, b), rep
            for k, (a, b) in enumerate(zip(outs, ref_out)):
                assert torch.equal(a, b), (rep, k)
        # a scalar-column job (cols % 4 != 0) sends the table down gptst_pool_jobs; results unchanged
        e7, p7 = torch.randn(50, 4, generator=g).to(dev), torch.randn(4, 2070, generator=g).to(dev)
        pj, outs = table(extra=[(e7, p7)])
        got = ops.mask_random(na, total, u24=True, jobs=pj)
        assert torch.equal(got, ref_mask[-1]) if not adaptive else True
        assert torch.allclose(outs[-1], e7 @ p7, rtol=1e-5, atol=1e-5)
        for a, b in zip(outs[:-1], ref_out):
            assert torch.equal(a, b)
    assert _handoff_timeouts() == 0


@pytest.mark.parametrize("ada_all", [1, 0])
def test_mask_adaptive_ties_and_caller_zeroed_scratch(ada_all):
    """Multi-launch adaptive generation with heavily tied noise in BOTH selections (the tie workgroup of selection A's mask write also has
    to feed selection R's first-digit histogram), scratch handed over zeroed (ws_zeroed: no zeroing launch), several budgets — bit-exact
    against the oracle; and the random selection with a caller-zeroed scratch."""
    from gptst_amd import ops, synth
    dev = _dev()
    B, T, N, HS = 8, 12, 170, 10
    M = B * T * N
    g = torch.Generator().manual_seed(77)
    label_ref = torch.randint(0, HS, (M,), generator=g)
    label = label_ref.to(torch.int32).to(dev)
    list_c = synth.class_order(HS, 9)
    na = torch.randint(0, 8, (M,), generator=g).float() / 8 + 0.0625
    nr = torch.randint(0, 8, (M,), generator=g).float() / 8 + 0.0625
    words = ops.mask_ws_floats()
    for total, frac in ((int(M * 0.25), 0.5), (int(M * 0.25), 0.93), (int(M * 0.6), 0.3), (37, 0.5)):
        ada = int(total * frac)
        rnd_n = total - ada
        # the oracle's sort is unstable on ties, as the reference's: its restatement with the kernel's documented tie rule (lowest index first)
        def drop(values, k):
            m = torch.ones(values.numel(), dtype=torch.int64)
            m[torch.sort(-values, stable=True)[1][:k]] = 0
            return m
        sel_c = torch.zeros(M, dtype=torch.int64)
        num, i = 0, 0
        while num < ada:
            sel_c[label_ref == list_c[i]] = 1
            num = int(sel_c.sum()); i += 1
        sel_d, dnum = torch.zeros(M, dtype=torch.int64), 0
        if ada_all and i >= 2:
            for k in range(i - 1):
                sel_d[label_ref == list_c[k]] = 1
            dnum = int(sel_d.sum())
            sel_f = (label_ref == list_c[i - 1]).long()
        else:
            sel_f = sel_c.clone()
        m_ada_r = drop(sel_f.float() * na, ada - dnum) * (1 - sel_d)
        m_rnd_r = drop(m_ada_r.float() * nr, rnd_n)
        fin_r = m_ada_r * m_rnd_r
        if len(torch.unique(na)) == M:                       # (never here) without ties the oracle itself must agree
            assert torch.equal(O.adaptive_mask(label_ref.view(B, T, N), list_c, na, nr, ada, rnd_n, "all" if ada_all else "half")[2], fin_r)
        with _mask_path(1):
            for ws in (None, torch.zeros(words, device=dev)):
                m_ada, m_rnd, mask = ops.mask_adaptive(label, None, torch.tensor(list_c, dtype=torch.int32, device=dev),
                                                       torch.tensor([ada, rnd_n], dtype=torch.int32, device=dev), na.to(dev), nr.to(dev),
                                                       ada_all, 1, ws=ws)
                assert torch.equal(m_ada.cpu().long(), m_ada_r), (total, frac, ws is None)
                assert torch.equal(m_rnd.cpu().long(), m_rnd_r), (total, frac, ws is None)
                assert torch.equal(mask.cpu().long(), fin_r.view(-1)), (total, frac, ws is None)
        for path, ws in ((0, None), (0, torch.zeros(words, device=dev)), (1, None), (1, torch.zeros(words, device=dev)), (2, None)):        # the lattice paths (0: the cooperative launch): the same tie rule
            with _mask_path(path):
                m_ada, m_rnd, mask = ops.mask_adaptive(label, None, torch.tensor(list_c, dtype=torch.int32, device=dev),
                                                       torch.tensor([ada, rnd_n], dtype=torch.int32, device=dev), na.to(dev), nr.to(dev),
                                                       ada_all, 1, ws=ws, u24=True)
            assert torch.equal(m_ada.cpu().long(), m_ada_r) and torch.equal(m_rnd.cpu().long(), m_rnd_r), ("u24", path, total, frac)
            assert torch.equal(mask.cpu().long(), fin_r.view(-1)), ("u24", path, total, frac)
    noise = synth.make_noise(M, 3)
    ref = O.random_mask(noise, 0.25)
    with _mask_path(1):
        got = ops.mask_random(noise.to(dev), int(M * 0.25), ws=torch.zeros(words, device=dev))
    assert torch.equal(got.cpu().to(torch.int64), ref)


# ---------------------------------------------------------------------------------------------------------------
def test_small_projections():
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    rows, C = 1000, 64
    for J, lda in ((1, 3), (2, 4), (10, 10), (40, 40)):
        a = rnd(rows, lda, g=g); W = rnd(C, J, g=g); b = rnd(C, g=g)
        mask = (torch.rand(rows, J, generator=g) > 0.3).float()
        Y = ops.lin_in(a.to(dev), lda, J, W.to(dev), b.to(dev), C)
        close(Y, a[:, :J] @ W.t() + b, what="lin_in")
        am = torch.where(mask != 0, a[:, :J], torch.full_like(a[:, :J], -1.5))
        Y = ops.lin_in(a.to(dev), lda, J, W.to(dev), b.to(dev), C, mask=mask.to(dev), fill=-1.5)
        close(Y, am @ W.t() + b, what="lin_in masked")
        Wt = W.t().contiguous()
        Y = ops.lin_in(a.to(dev), lda, J, Wt.to(dev), None, C, wlayout=1)
        close(Y, a[:, :J] @ W.t(), what="lin_in wlayout1")
        X = rnd(rows, C, g=g); W2 = rnd(J, C, g=g); b2 = rnd(J, g=g)
        close(ops.rowdot(X.to(dev), W2.to(dev), b2.to(dev)), X @ W2.t() + b2, what="rowdot")
        close(ops.rowdot(X.to(dev), W2.to(dev), b2.to(dev), softmax=True), torch.softmax(X @ W2.t() + b2, -1), what="rowdot softmax")
        Zs, lab = ops.rowdot(X.to(dev), W2.to(dev), b2.to(dev), softmax=True, want_label=True)
        assert torch.equal(lab.cpu().long(), torch.sort(Zs.cpu(), dim=-1, descending=True, stable=True)[1][..., 0]), "rowdot label = first argmax"
        out0 = torch.zeros(C, J, device=dev); out1 = torch.zeros(J, C, device=dev)
        cs = torch.zeros(C, device=dev); asum = torch.zeros(J, device=dev)
        ops.rowouter(a.to(dev), lda, J, X.to(dev), out0, 0, csum=cs, asum=asum, mask=mask.to(dev), fill=-1.5)
        ops.rowouter(a.to(dev), lda, J, X.to(dev), out1, 1)
        close(out0, X.t() @ am, what="rowouter layout0 masked")
        close(out1, a[:, :J].t() @ X, what="rowouter layout1")
        close(cs, X.sum(0), what="csum"); close(asum, am.sum(0), what="asum")
    cs = torch.zeros(C, device=dev)
    ops.rowouter(None, 0, 0, X.to(dev), None, 0, csum=cs)
    close(cs, X.sum(0), what="csum only")


@pytest.mark.parametrize("E,K,rows", [(16, 1, 384), (4, 1, 24), (4, 12, 32), (8, 1, 130), (4, 12, 2)])
def test_timefeat(E, K, rows):
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(E + K)
    names = ["ln_day", "ln_week", "ln1", "ln2", "ln"]
    sd = {}
    for n in names:
        i = K if n in ("ln_day", "ln_week") else E
        sd["t.%s.weight" % n] = rnd(E, i, g=g, scale=0.5).requires_grad_()
        sd["t.%s.bias" % n] = rnd(E, g=g, scale=0.5).requires_grad_()
    if K == 1:
        tidx = rnd(rows // 12 if rows % 12 == 0 else rows, 12 if rows % 12 == 0 else 1, 2, g=g)
        ref = O.time_feature(sd, "t.", tidx).reshape(rows, E)
    else:
        tidx = rnd(rows, 12, 2, g=g)
        ref = O.time_feature_spg(sd, "t.", tidx)
    go = rnd(rows, E, g=g)
    go = go * (ref.detach().abs() > 1e-5)       # a pre-activation within fp32 noise of 0 may pick the other LReLU slope: not a parity question
    (ref * go).sum().backward()
    params = []
    for n in names:
        params += [sd["t.%s.weight" % n].detach().to(dev).contiguous(), sd["t.%s.bias" % n].detach().to(dev).contiguous()]
    out = ops.timefeat_fwd(params, tidx.to(dev).contiguous(), rows, K)
    close(out, ref, what="timefeat out")
    grads = [torch.zeros_like(p) for p in params]
    ops.timefeat_bwd(params, grads, tidx.to(dev).contiguous(), go.to(dev), rows, K)
    i = 0
    for n in names:
        close(grads[i], sd["t.%s.weight" % n].grad, what="timefeat d%s.w" % n)
        close(grads[i + 1], sd["t.%s.bias" % n].grad, what="timefeat d%s.b" % n)
        i += 2


def test_loss_and_kl():
    from gptst_amd import ops, synth
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    B, T, N, HS = 3, 12, 20, 5
    for base, thresh in ((1, 0.0), (2, 0.001)):
        rows = B * T * N
        src = synth.make_batch(B, T, N, base, seed=3)
        out = rnd(B, T, N, base, g=g).requires_grad_()
        vis = (torch.rand(B, T, N, base, generator=g) > 0.25).float()          # 1 = visible
        loss = O.mae_loss(out, src[..., :base], (1 - vis).long(), synth.SCALER_MEAN, synth.SCALER_STD, thresh)
        loss.backward()
        stats = torch.zeros(8, device=dev)
        o, s, m = out.detach().to(dev).contiguous(), src.to(dev).contiguous(), vis.to(dev).contiguous()
        ops.mae_fwd(o, s, base + 2, m, synth.SCALER_STD, synth.SCALER_MEAN, thresh, rows, base, stats)
        st = stats.cpu()
        assert abs(float(st[0] / st[1]) - float(loss)) < 1e-4 * float(loss)
        dO = ops.mae_bwd(o, s, base + 2, m, synth.SCALER_STD, synth.SCALER_MEAN, thresh, rows, base, stats)
        close(dO, out.grad, what="mae bwd")
    logits = rnd(rows, HS, g=g).requires_grad_()
    prob = torch.softmax(logits, -1)
    c = torch.softmax(rnd(B * T, HS, N, g=g), 1)                                # (BT, HS, N)
    eb = c.view(B, T, HS, N).transpose(-1, -2).reshape(rows, HS)
    ls = torch.nn.functional.kl_div(prob.log(), eb, reduction="sum") * 0.1
    ls.backward()
    stats = torch.zeros(8, device=dev)
    dl = ops.kl(prob.detach().to(dev).contiguous(), c.to(dev).contiguous(), N, 0.1, stats)
    assert abs(float(stats[2].cpu()) * 0.1 - float(ls)) < 1e-4 * abs(float(ls))
    close(dl, logits.grad, what="kl dlogit")


def test_clip_adam_matches_torch():
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    nA, nB, nC = 5000, 700, 100
    n = nA + nB + nC
    p0 = rnd(n, g=g)
    pa = p0[:nA].clone().requires_grad_(); pb = p0[nA:nA + nB].clone().requires_grad_()
    opt = torch.optim.Adam([pa, pb], lr=0.003, eps=1e-8)
    p = p0.to(dev).clone(); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    tA = tB = 0
    for step in range(6):
        actB = step >= 3
        gr = rnd(n, g=g, scale=3.0 if step % 2 else 0.01)
        pa.grad = gr[:nA].clone()
        pb.grad = gr[nA:nA + nB].clone() if actB else None
        torch.nn.utils.clip_grad_norm_([pa, pb], 5)
        opt.step()
        tA += 1
        tB += 1 if actB else 0
        hy = torch.zeros(16)
        hy[0] = 0.003 / (1 - 0.9 ** tA); hy[1] = (1 - 0.999 ** tA) ** 0.5
        if tB:
            hy[2] = 0.003 / (1 - 0.9 ** tB); hy[3] = (1 - 0.999 ** tB) ** 0.5
        hy[4], hy[5], hy[6], hy[7], hy[8], hy[9], hy[10] = 0.9, 0.999, 1e-8, 5.0, float(actB), 0.0, 1.0
        hy[11], hy[12] = 1 - 0.9, 1 - 0.999               # 1 - beta as the host rounds it (what torch.optim.Adam passes to its kernels)
        stats = torch.zeros(8, device=dev)
        prev = p.clone()
        ops.clip_adam(p, gr.to(dev), m, v, nA, nB, hy.to(dev), stats)
        ref = torch.cat([pa.detach(), pb.detach(), p0[nA + nB:]])
        close(p, ref, tol=2e-6, what="adam step %d" % step)
        # the LENGTH of the update (r04): 1.f - 0.999f in the kernel made every step 6.4e-6 too long — 2e-8 per element, below the fp32
        # spacing of p ~ 1 and invisible above, but systematic: it put the weights 300x further from an fp64 trajectory than the fp32 oracle
        # is (tests/test_gpu_step.py).  The signed error along the update direction averages the rounding noise out.
        sel = torch.arange(n, device=dev) < (nA + (nB if actB else 0))
        d = ((p - ref.to(dev)) * torch.sign(ref.to(dev) - prev))[sel].double()
        upd = float((ref.to(dev) - prev)[sel].abs().double().mean())
        assert abs(float(d.mean())) < 1e-6 * upd + 3e-9, (step, float(d.mean()), upd)


@pytest.mark.parametrize("det", [0, 1])
def test_pool_jobs_mixed_table(det):
    """gptst_pool_jobs: problems of all three kinds, each with its own embedding / shape, in one call (60 jobs -> 2 launches);
    det = 1: the deterministic routing (one single-owner launch per distinct demb)."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    J = ops.PoolJobs()
    checks = []
    shapes = [(384, 16, 4096), (170, 16, 64), (32, 4, 1920), (384, 4, 1700), (24, 8, 70), (1, 1, 4160), (207, 16, 2070), (5, 3, 7)]
    for rep in range(3):
        for (R, K, cols) in shapes:
            emb, pool = rnd(R, K, g=g), rnd(K, cols, g=g)
            out = J.fwd(emb.to(dev), pool.to(dev))
            checks.append((out, emb @ pool, "fwd %s" % ((R, K, cols),)))
            for ns in ((1, 3) if rep == 0 else (1,)):
                dW = rnd(ns * R, cols, g=g)
                dpool = rnd(K, cols, g=g)
                dp = dpool.to(dev)
                J.bwd_pool(emb.to(dev), dW.to(dev), dp, nsplit=ns)
                checks.append((dp, dpool + emb.t() @ dW.view(ns, R, cols).sum(0), "bwd_pool %s ns=%d" % ((R, K, cols), ns)))
                demb = rnd(R, K, g=g)
                de = demb.to(dev)
                J.bwd_emb(dW.to(dev), pool.to(dev), de, nsplit=ns)
                J.bwd_emb(dW.to(dev), pool.to(dev), de, nsplit=ns)                       # two jobs add into one demb
                checks.append((de, demb + 2 * dW.view(ns, R, cols).sum(0) @ pool.t(), "bwd_emb %s ns=%d" % ((R, K, cols), ns)))
    assert len(J.jobs) > 96
    ops.set_deterministic(det)
    try:
        J.launch()
    finally:
        ops.set_deterministic(0)
    for got, ref, what in checks:
        close(got, ref, what=what)


@pytest.mark.parametrize("N,d,Hm", [(170, 16, 8), (33, 4, 5), (1, 1, 1), (300, 8, 16)])
def test_pool_jobs_gram_kind_is_bitwise_gram_of_the_generated_factor(N, d, Hm):
    """Job kind 3 of gptst_pool_jobs: G_n = A_n^T A_n straight from (node embedding, hyperedge pool), in the SAME launch as the job that
    materialises A — bit-identical to gptst_gram_fwd of that job's output (same fmaf orders), and close to the fp64 product."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(N + d)
    ne, adj = rnd(N, d, g=g), rnd(d, Hm, 12, g=g)
    J = ops.PoolJobs()
    A = J.fwd(ne.to(dev), adj.to(dev).view(d, Hm * 12))
    G = J.gram(ne.to(dev), adj.to(dev).view(d, Hm * 12), torch.empty(N, 12, 12, device=dev))
    J.launch()
    assert torch.equal(G, ops.gram_fwd(A.view(N, Hm, 12)))
    Ad = torch.einsum("nd,dht->nht", ne.double(), adj.double())
    close(G, torch.einsum("nht,nhu->ntu", Ad, Ad).float(), what="gram job")


def test_timefeat_jobs_equal_single_launches():
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, T = 5, 12
    tidx = rnd(B, T, 2, g=g).to(dev)

    def params(E, K):
        return [t.to(dev) for t in (rnd(E, K, g=g), rnd(E, g=g), rnd(E, K, g=g), rnd(E, g=g), rnd(E, E, g=g), rnd(E, g=g), rnd(E, E, g=g),
                                    rnd(E, g=g), rnd(E, E, g=g), rnd(E, g=g))]
    jobs = [(params(16, 1), B * T, 1), (params(4, 1), B * T, 1), (params(4, 12), B, 12), (params(8, 1), B * T, 1), (params(2, 12), B, 12)]
    outs = ops.timefeat_jobs_fwd(jobs, tidx)
    bj = []
    for (pp, rows, K), o in zip(jobs, outs):
        ref = ops.timefeat_fwd(pp, tidx, rows, K)
        assert torch.equal(o, ref)
        go = rnd(rows, pp[1].numel(), g=g).to(dev)
        g1 = [torch.zeros_like(t) for t in pp]
        ops.timefeat_bwd(pp, g1, tidx, go, rows, K)
        g2 = [torch.zeros_like(t) for t in pp]
        bj.append((pp, g2, go, rows, K, g1))
    ops.timefeat_jobs_bwd([j[:5] for j in bj], tidx)
    for pp, g2, go, rows, K, g1 in bj:
        for a, b in zip(g1, g2):
            close(b, a.cpu(), tol=1e-5, what="timefeat_jobs bwd")
    # deterministic variant: one workgroup per job walks its row blocks in order; two runs are bit-identical
    runs = []
    for rep in range(2):
        g3 = [[torch.zeros_like(t) for t in j[0]] for j in bj]
        ops.set_deterministic(1)
        try:
            ops.timefeat_jobs_bwd([(j[0], g3[q], j[2], j[3], j[4]) for q, j in enumerate(bj)], tidx)
        finally:
            ops.set_deterministic(0)
        runs.append(g3)
    for q, j in enumerate(bj):
        for a, b, c in zip(j[5], runs[0][q], runs[1][q]):
            close(b, a.cpu(), tol=1e-5, what="timefeat_jobs bwd det")
            assert torch.equal(b, c)


@pytest.mark.parametrize("J,rows,C", [(1, 65280, 64), (2, 1000, 64), (1, 37, 64), (1, 3000, 128), (2, 301, 128)])
def test_tail_mae_matches_unfused_ops(J, rows, C):
    """tails.hip mae tail == rowdot + mae_fwd + mae_bwd(normalize=False) + lin_in + rowouter, and the statistics of the oracle loss."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    lda = J + 2
    dec, W, b = rnd(rows, C, g=g), rnd(J, C, g=g, scale=0.2), rnd(J, g=g)
    src = rnd(rows, lda, g=g)
    mask = (torch.rand(rows * J, generator=g) > 0.25).float()
    sigma, mu, thr = 146.0, 230.0, 0.0
    stats, sws = torch.zeros(8, device=dev), ops.tail_sws(rows, dev)
    out, dd, part = ops.tail_mae(dec.to(dev), W.to(dev), b.to(dev), src.to(dev), lda, mask.to(dev), sigma, mu, thr, sws)
    ops.stats_fold(sws, stats)
    out_r = dec @ W.t() + b
    close(out, out_r, what="tail out")
    M = (1 - mask).view(rows, J)
    p_, y_ = (out_r * sigma + mu) * M, (src[:, :J] * sigma + mu) * M
    keep = y_ > thr
    assert abs(float(stats[1]) - float(keep.sum())) < 0.5
    assert abs(float(stats[0]) - float((y_ - p_).abs()[keep].sum())) < 1e-4 * float((y_ - p_).abs()[keep].sum())
    a = torch.sign(p_ - y_) * M * sigma * keep
    close(dd, a @ W, what="tail d_dec")
    gwb = part.sum(0).cpu()
    close(gwb[:J * C].view(J, C), a.t() @ dec, what="tail gW")
    close(gwb[J * C:], a.sum(0), what="tail gb")


@pytest.mark.parametrize("HS,N,BT,C", [(10, 170, 48, 64), (5, 20, 7, 64), (16, 33, 5, 64), (10, 301, 12, 128), (3, 17, 2, 128)])
def test_tail_kl_matches_unfused_ops(HS, N, BT, C):
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(22)
    rows = BT * N
    h2, W3 = rnd(rows, C, g=g), rnd(HS, C, g=g, scale=0.2)
    prob = torch.softmax(rnd(rows, HS, g=g), -1)
    c = torch.softmax(rnd(BT, HS, N, g=g), 1).contiguous()
    stats, sws = torch.zeros(8, device=dev), ops.tail_sws(rows, dev)
    dh2, part = ops.tail_kl(h2.to(dev), W3.to(dev), prob.to(dev), c.to(dev), N, 0.1, sws)
    ops.stats_fold(sws, stats)
    stats_r = torch.zeros(8, device=dev)
    dlogit = ops.kl(prob.to(dev), c.to(dev), N, 0.1, stats_r).cpu()
    assert abs(float(stats[2]) - float(stats_r[2])) < 1e-4 * abs(float(stats_r[2])) + 1e-6
    eb = c.permute(0, 2, 1).reshape(rows, HS)
    close(dlogit, 0.1 * (prob * eb.sum(-1, keepdim=True) - eb), what="kl dlogit")
    close(dh2, dlogit @ W3, what="tail d_h2")
    gwb = part.sum(0).cpu()
    close(gwb[:HS * C].view(HS, C), dlogit.t() @ h2, what="tail gW3")
    close(gwb[HS * C:], dlogit.sum(0), what="tail gb3")


@pytest.mark.parametrize("BT,N,C", [(24, 37, 64), (6, 301, 128)])
def test_wgrad_colsum_of_dpre_and_column_window_jobs(BT, N, C):
    """gptst_wgrad_colsum(which=2): rows [dW | column sums of dPre] (hyperTem's weight AND bias gradient from one pass), consumed by
    pool jobs that read column windows of those rows (ldx > cols)."""
    from gptst_amd import ops
    from gptst_amd.ops import MODE_TIME, PRO_DPRE
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    d = 8
    A, D, Y = rnd(BT * N, C, g=g), rnd(BT * N, C, g=g), rnd(BT * N, C, g=g)
    dpre = D * torch.where(Y > 0, torch.ones_like(Y), torch.full_like(Y, 0.01))
    dWb, ns = ops.wgrad(A.to(dev), D.to(dev), MODE_TIME, BT, N, D2=Y.to(dev), pro=PRO_DPRE, colsum_d=True)
    got = dWb.view(ns, BT, C * C + C).sum(0).cpu()
    dW_ref = torch.einsum("gmi,gmo->gio", A.view(BT, N, C), dpre.view(BT, N, C)).reshape(BT, C * C)
    close(got[:, :C * C], dW_ref, what="wgrad dW")
    close(got[:, C * C:], dpre.view(BT, N, C).sum(1), what="wgrad colsum dPre")
    te, wpool, bpool = rnd(BT, d, g=g), rnd(d, C * C, g=g), rnd(d, C, g=g)
    gw, gb, dte = torch.zeros(d, C * C, device=dev), torch.zeros(d, C, device=dev), torch.zeros(BT, d, device=dev)
    J = ops.PoolJobs()
    J.bwd_pool(te.to(dev), dWb[:, :C * C], gw, nsplit=ns)
    J.bwd_pool(te.to(dev), dWb[:, C * C:], gb, nsplit=ns)
    J.bwd_emb(dWb[:, :C * C], wpool.to(dev), dte, nsplit=ns)
    J.bwd_emb(dWb[:, C * C:], bpool.to(dev), dte, nsplit=ns)
    J.launch()
    db_ref = dpre.view(BT, N, C).sum(1)
    close(gw, te.t() @ dW_ref, what="window bwd_pool W")
    close(gb, te.t() @ db_ref, what="window bwd_pool b")
    close(dte, dW_ref @ wpool.t() + db_ref @ bpool.t(), what="window bwd_emb")


@pytest.mark.parametrize("mode,BT,N", [(1, 384, 170), (0, 384, 170), (1, 24, 37), (0, 24, 37), (1, 7, 5)])
def test_apply_wgrad_fused_layer_backward(mode, BT, N):
    """gptst_apply_wgrad == the reference backward of out = lrelu(S W_g + b_g + x): dS, dW_g, db_g (row-split partials summed)."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(41 + mode)
    C = 64
    G = BT if mode == 0 else N
    dO, Y, S = rnd(BT, N, C, g=g), rnd(BT, N, C, g=g), rnd(BT, N, C, g=g)
    W = rnd(G, C, C, g=g, scale=0.2)
    dpre = dO * torch.where(Y > 0, torch.ones_like(Y), torch.full_like(Y, 0.01))
    if mode == 0:
        dS_r = torch.einsum("gno,gio->gni", dpre, W)
        dW_r = torch.einsum("gni,gno->gio", S, dpre)
        db_r = dpre.sum(1)
    else:
        dS_r = torch.einsum("bno,nio->bni", dpre, W)
        dW_r = torch.einsum("bni,bno->nio", S, dpre)
        db_r = dpre.sum(0)
    dS, dW, db, ns = ops.apply_wgrad(dO.view(-1, C).to(dev), Y.view(-1, C).to(dev), S.view(-1, C).to(dev), W.to(dev), mode, BT, N)
    close(dS.view(BT, N, C), dS_r, what="apply_wgrad dS")
    close(dW.view(ns, G, C, C).sum(0), dW_r, what="apply_wgrad dW")
    close(db.view(ns, G, C).sum(0), db_r, what="apply_wgrad db")


@pytest.mark.parametrize("rows", [65280, 1000, 37])
def test_linear_bwd_fused(rows):
    """gptst_linear_bwd == dX = dY Wp + dOut*lrelu'(out), dWp = dY^T X, dbp = colsum(dY) (cap's entry Linear + residual branch)."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(51)
    C = 64
    dY, X, dO, out = rnd(rows, C, g=g), rnd(rows, C, g=g), rnd(rows, C, g=g), rnd(rows, C, g=g)
    Wp = rnd(C, C, g=g, scale=0.2)
    dX, dWp, dbp, ns = ops.linear_bwd(dY.to(dev), X.to(dev), Wp.to(dev), dO.to(dev), out.to(dev))
    close(dX, dY @ Wp + dO * torch.where(out > 0, torch.ones_like(out), torch.full_like(out, 0.01)), what="linear_bwd dX")
    close(dWp.view(ns, C, C).sum(0), dY.t() @ X, what="linear_bwd dWp")
    close(dbp.sum(0), dY.sum(0), what="linear_bwd dbp")


@pytest.mark.parametrize("R,K,cols", [(384, 16, 4096), (170, 16, 4096), (384, 4, 1700), (37, 5, 260), (20, 16, 64), (1, 3, 8)])
def test_poolgen_mfma_bitwise(R, K, cols):
    """The MFMA forward of the pool jobs reproduces the VALU forward BIT FOR BIT (an fp32 MFMA is the fmaf chain over its k values in
    order), so switching it on changes no number anywhere downstream — and both match the fp64 product to fp32 accuracy."""
    from gptst_amd import _C, ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    emb, pool = rnd(R, K, g=g).to(dev), rnd(K, cols, g=g).to(dev)
    outs = []
    for mfma in (1, 0):
        _C.lib().call("gptst_tune", 10, mfma)
        try:
            J = ops.PoolJobs()
            o = J.fwd(emb, pool)
            J.launch()
            outs.append(o.clone())
        finally:
            _C.lib().call("gptst_tune", 10, 1)
    assert torch.equal(outs[0], outs[1])
    close(outs[0], emb.double().cpu() @ pool.double().cpu(), what="poolgen fwd")


@pytest.mark.parametrize("B,N", [(32, 170), (3, 37), (2, 16), (9, 5)])
def test_hypertem_bwd_wgrad_one_launch_equals_two(B, N):
    """gptst_hypertem_bwd_wgrad == gptst_hypertem_bwd + gptst_wgrad_colsum: dX, dG and the weight gradients bit for bit (same device code in
    one grid); the bias column sums agree to the last bits (the two compilations of the shared body round a handful of entries 1 ulp
    apart, both equally close to the fp64 sums) and are checked against fp64 as well."""
    from gptst_amd import ops
    from gptst_amd.ops import MODE_TIME, PRO_DPRE
    dev = _dev()
    g = torch.Generator().manual_seed(41)
    C, T = 64, 12
    X, dO = rnd(B, T, N, C, g=g).to(dev), rnd(B, T, N, C, g=g).to(dev)
    G = (rnd(N, T, T, g=g) * 0.1).to(dev)
    Wbt, bbt = (rnd(B * T, C, C, g=g) * 0.1).to(dev), rnd(B * T, C, g=g).to(dev)
    R, out = ops.hypertem_fwd(X, G, Wbt, bbt)
    dx1, _, dG1 = ops.hypertem_bwd(dO, out, X, G, Wbt, want_dbias=False)
    dWb1, ns1 = ops.wgrad(R.view(-1, C), dO.view(-1, C), MODE_TIME, B * T, N, D2=out.view(-1, C), pro=PRO_DPRE, colsum_d=True)
    dx2, dWb2, ns2, dG2 = ops.hypertem_bwd_wgrad(dO, out, X, G, Wbt, R)
    assert ns1 == ns2
    assert torch.equal(dx1, dx2) and torch.equal(dG1, dG2) and torch.equal(dWb1[:, :C * C], dWb2[:, :C * C])
    db_ref = (dO * torch.where(out > 0, 1.0, 0.01)).view(B * T, N, C).double().sum(1).cpu()
    close(dWb2[:, C * C:], db_ref, what="fused db")
    assert float((dWb1[:, C * C:] - dWb2[:, C * C:]).abs().max()) <= 4e-7 * float(db_ref.abs().max())


def _lg(t):
    return torch.where(t > 0, 1.0, 0.01)


@pytest.mark.parametrize("B,N", [(32, 170), (3, 37), (2, 16), (9, 5)])
def test_hypertem_bwd_dpre_chain_and_rebuilt_R(B, N):
    """The dPre-chain forms of the fused hyperTem backward (Y = NULL: the incoming gradient already is dOut*lrelu'(out); premul: dX leaves
    multiplied by lrelu'(X)) and the weight gradient with R rebuilt from X (R = NULL) reproduce the legacy launch: dG / dW bit for bit
    (same arithmetic, R summed in the forward's order), dX up to the one extra rounding of the premultiplication."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(43)
    C, T = 64, 12
    X, dO = rnd(B, T, N, C, g=g).to(dev), rnd(B, T, N, C, g=g).to(dev)
    G = (rnd(N, T, T, g=g) * 0.1).to(dev)
    Wbt, bbt = (rnd(B * T, C, C, g=g) * 0.1).to(dev), rnd(B * T, C, g=g).to(dev)
    R, out = ops.hypertem_fwd(X, G, Wbt, bbt)
    R0, out0 = ops.hypertem_fwd(X, G, Wbt, bbt, want_R=False)
    assert R0 is None and torch.equal(out, out0)
    dx1, dWb1, ns, dG1 = ops.hypertem_bwd_wgrad(dO, out, X, G, Wbt, R)
    dpre = dO * _lg(out)
    for kw in (dict(R=R), dict(R=None)):                      # Y = NULL, with the saved and with the rebuilt R
        dx2, dWb2, ns2, dG2 = ops.hypertem_bwd_wgrad(dpre, None, X, G, Wbt, kw["R"])
        assert ns2 == ns and torch.equal(dx1, dx2) and torch.equal(dG1, dG2)
        assert torch.equal(dWb1[:, :C * C], dWb2[:, :C * C]), kw
        assert float((dWb1[:, C * C:] - dWb2[:, C * C:]).abs().max()) <= 4e-7 * float(dWb1[:, C * C:].abs().max())
    dx3, dWb3, _, dG3 = ops.hypertem_bwd_wgrad(dO, out, X, G, Wbt, None)          # legacy sign operand + rebuilt R
    assert torch.equal(dx1, dx3) and torch.equal(dG1, dG3) and torch.equal(dWb1[:, :C * C], dWb3[:, :C * C])
    dx4, dWb4, _, dG4 = ops.hypertem_bwd_wgrad(dpre, None, X, G, Wbt, None, premul=True)
    assert torch.equal(dG1, dG4) and torch.equal(dWb1[:, :C * C], dWb4[:, :C * C])
    assert torch.equal(dx4, dx1 * _lg(X))
    dx5, _, dG5 = ops.hypertem_bwd(dpre, None, X, G, Wbt, want_dbias=False, premul=True)
    assert torch.equal(dx5, dx4) and torch.equal(dG5, dG1)
    with pytest.raises(Exception):
        ops.hypertem_bwd_wgrad(dO, out, X, G, Wbt, R, premul=True)               # sign operand AND premultiplication: rejected


@pytest.mark.parametrize("mode,BT,N", [(1, 384, 170), (0, 384, 170), (1, 24, 37), (0, 36, 20)])
def test_apply_wgrad_dpre_chain(mode, BT, N):
    """gptst_apply_wgrad with Y = NULL (incoming gradient already dPre) and premul (dS * lrelu'(S)) against the legacy call."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(47)
    C = 64
    G_ = BT if mode == 0 else N
    dO, Y, S = (rnd(BT * N, C, g=g).to(dev) for _ in range(3))
    W = (rnd(G_, C, C, g=g) * 0.1).to(dev)
    dS1, dW1, db1, ns = ops.apply_wgrad(dO, Y, S, W, mode, BT, N)
    dpre = dO * _lg(Y)
    dS2, dW2, db2, _ = ops.apply_wgrad(dpre, None, S, W, mode, BT, N)
    assert torch.equal(dS1, dS2) and torch.equal(dW1, dW2) and torch.equal(db1, db2)
    dS3, dW3, db3, _ = ops.apply_wgrad(dpre, None, S, W, mode, BT, N, premul=True)
    assert torch.equal(dS3, dS1 * _lg(S)) and torch.equal(dW1, dW3) and torch.equal(db1, db3)


@pytest.mark.parametrize("rows", [65280, 1000, 16])
def test_linear_bwd_dpre_chain(rows):
    """gptst_linear_bwd with out = NULL: dX = dY Wp + dPre, and with premul (dY Wp + dPre) * lrelu'(X), against the legacy call."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(53)
    C = 64
    dY, X, dO, out = (rnd(rows, C, g=g).to(dev) for _ in range(4))
    Wp = (rnd(C, C, g=g) * 0.1).to(dev)
    dX1, dWp1, dbp1, ns = ops.linear_bwd(dY, X, Wp, dO, out)
    dpre = dO * _lg(out)
    dX2, dWp2, dbp2, _ = ops.linear_bwd(dY, X, Wp, dpre, None)
    close(dX2, dX1.cpu(), what="linear_bwd chain dX")          # fmaf(dOut, lrelu', acc) vs acc + dPre: one rounding apart
    assert torch.equal(dWp1, dWp2) and torch.equal(dbp1, dbp2)
    dX3, dWp3, _, _ = ops.linear_bwd(dY, X, Wp, dpre, None, premul=True)
    assert torch.equal(dX3, dX2 * _lg(X)) and torch.equal(dWp1, dWp3)


def test_tails_premul():
    """tail_mae / tail_kl with premul: the data gradient leaves multiplied by lrelu'(input activation); everything else unchanged."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(59)
    B, T, N, C, J, HS = 2, 12, 37, 64, 1, 6
    rows = B * T * N
    dec = rnd(rows, C, g=g).to(dev)
    W, b = rnd(J, C, g=g).to(dev), rnd(J, g=g).to(dev)
    src = rnd(B, T, N, J + 2, g=g).to(dev)
    mask = (torch.rand(rows * J, generator=g) > 0.3).float().to(dev)
    res = []
    for pm in (False, True):
        sws = ops.tail_sws(rows, dev)
        res.append(ops.tail_mae(dec, W, b, src, J + 2, mask, 146.0, 230.0, 0.0, sws, premul=pm) + (sws,))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
    assert torch.equal(res[1][1], res[0][1] * _lg(dec))
    W3 = rnd(HS, C, g=g).to(dev)
    prob = torch.softmax(rnd(rows, HS, g=g), -1).to(dev)
    c = torch.softmax(rnd(B * T, HS, N, g=g), 1).to(dev)
    res = []
    for pm in (False, True):
        sws = ops.tail_sws(rows, dev)
        res.append(ops.tail_kl(dec, W3, prob, c, N, 0.1, sws, premul=pm) + (sws,))
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert torch.equal(res[1][0], res[0][0] * _lg(dec))


def _philox_ref(i, step, seed):
    """Philox4x32-10 (Salmon et al. 2011) in plain Python: counter (i_lo, i_hi, step, 0), key (seed, 0x5EED) -> four uniforms in [0,1)"""
    M0, M1, W0, W1, mask = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    c, k = [i & mask, (i >> 32) & mask, step, 0], [seed, 0x5EED]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & mask, p1 & mask, ((p0 >> 32) ^ c[3] ^ k[1]) & mask, p0 & mask]
        k = [(k[0] + W0) & mask, (k[1] + W1) & mask]
    return [(w >> 8) * 2.0 ** -24 for w in c]


@pytest.mark.parametrize("B,N,HS,HT", [(32, 170, 10, 16), (2, 20, 5, 6), (3, 37, 16, 5), (1, 266, 20, 16)])
def test_cap_cross_folded_into_its_neighbours(B, N, HS, HT):
    """gptst_cap_cross_rec_fwd == gptst_cap_cross_fwd + gptst_cap_rec_fwd bit for bit, and gptst_cap_cross_route_bwd ==
    gptst_cap_cross_bwd + gptst_cap_route_bwd (same arithmetic; dS never leaves LDS)."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(61)
    C, T = 64, 12
    s = rnd(B * T, HS, C, g=g).to(dev)
    dyn = (rnd(B, HT, T * HS, g=g) * 0.3).to(dev)
    tmpl = (torch.arange(1, T + 1) / 12.0).float().to(dev)
    c = torch.softmax(rnd(B * T, HS, N, g=g), 1).to(dev).contiguous()
    v1, Ht1, Rt1 = ops.cap_cross_fwd(s, dyn, tmpl, B, T, HS, HT)
    rec1 = ops.cap_rec_fwd(c, v1, N, C)
    fused = ops.cap_cross_rec_fwd(s, dyn, tmpl, c, B, T, N, HS, HT)
    assert fused is not None or HS * T > 200                 # (HS = 20: 240 tokens + c[bt] exceed the 80 KB the fused launch allows itself)
    if fused is not None:
        v2, Ht2, Rt2, rec2 = fused
        assert torch.equal(v1, v2) and torch.equal(Ht1, Ht2) and torch.equal(Rt1, Rt2) and torch.equal(rec1, rec2)
    X = rnd(B, T, N, C, g=g).to(dev)
    Wp, bp = (rnd(C, C, g=g) * 0.1).to(dev), rnd(C, g=g).to(dev)
    dc1, dv = rnd(B * T, HS, N, g=g).to(dev), rnd(B * T, HS, C, g=g).to(dev)
    dS, ddyn1 = ops.cap_cross_bwd(dv, s, Rt1, Ht1, dyn, tmpl, B, T, HS, HT)
    dY1, dl1 = ops.cap_route_bwd(X, Wp, bp, c, dc1, dS)
    fb = ops.cap_cross_route_bwd(X, Wp, bp, c, dc1, dv, s, Rt1, Ht1, dyn, tmpl, B, T, HS, HT)
    if fb is None:            # the prologue's scratch (T*HS tokens) must fit the capsule tile of N nodes: small N with many clusters keeps two launches
        assert (T * HS + 2 * HT + HS) * (C + 4) + HT * T * HS > ((N + 15) // 16 * 16) * (C + 4) + max(C * C, 2 * 16 * (((N + 3) // 4 * 4) | 1))
        return
    dY2, dl2, ddyn2 = fb
    close(ddyn2, ddyn1.cpu(), tol=2e-6, what="folded ddyn")
    close(dY2, dY1.cpu(), tol=2e-6, what="folded dY")
    close(dl2, dl1.cpu(), tol=2e-6, what="folded dlogit")
    # r04: the cross-time backward as a ROLE of the launch (B extra workgroups publish dS per sample: write-through stores + a flag; the routing
    # workgroups wait for it behind their capsule GEMM) — the same arithmetic: bit-identical to the replicated prologue, call after call
    for rep in range(3):
        fr = ops.cap_cross_route_bwd(X, Wp, bp, c, dc1, dv, s, Rt1, Ht1, dyn, tmpl, B, T, HS, HT, flags=torch.zeros(4 * B, device=dev))
        assert fr is not None
        for a, b_, nm in zip(fr, fb, ("dY", "dlogit", "ddyn")):
            assert torch.equal(a, b_), "role form differs from the prologue form in %s (rep %d)" % (nm, rep)


@pytest.mark.parametrize("mode", ["out", "dpre", "dpre_premul"])
@pytest.mark.parametrize("B,N,HS,HT", [(32, 170, 10, 16), (2, 50, 5, 6), (3, 37, 16, 5), (2, 207, 10, 16)])
def test_cap_route_lin_bwd_equals_route_bwd_plus_linear_bwd(B, N, HS, HT, mode):
    """r05: gptst_cap_cross_route_lin_bwd == gptst_cap_cross_route_bwd + gptst_linear_bwd.  dX runs the same MFMA chain on the same operands
    (bit-identical); the weight / bias gradient comes as ONE partial per (b,t) summed over its nodes in order (510 row splits folded over four waves
    before): compared after the reduction, 2e-6 of the tensor scale."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(73)
    C, T = 64, 12
    s = rnd(B * T, HS, C, g=g).to(dev)
    dyn = (rnd(B, HT, T * HS, g=g) * 0.3).to(dev)
    tmpl = (torch.arange(1, T + 1) / 12.0).float().to(dev)
    c = torch.softmax(rnd(B * T, HS, N, g=g), 1).to(dev).contiguous()
    _, Ht, Rt = ops.cap_cross_fwd(s, dyn, tmpl, B, T, HS, HT)
    X = rnd(B, T, N, C, g=g).to(dev)
    Wp, bp = (rnd(C, C, g=g) * 0.1).to(dev), rnd(C, g=g).to(dev)
    dc1, dv = rnd(B * T, HS, N, g=g).to(dev), rnd(B * T, HS, C, g=g).to(dev)
    dout, out = rnd(B * T * N, C, g=g).to(dev), rnd(B * T * N, C, g=g).to(dev)
    ref = ops.cap_cross_route_bwd(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, B, T, HS, HT, flags=torch.zeros(4 * B, device=dev))
    if ref is None:
        pytest.skip("the one-launch routing backward does not serve this shape")
    dY, dl1, ddyn1 = ref
    o_, pm = (out, False) if mode == "out" else (None, mode == "dpre_premul")
    dX1, dWp1, dbp1, ns = ops.linear_bwd(dY, X.view(-1, C), Wp, dout, o_, premul=pm)
    # r06: the last nsplit (b,t) as two node halves with a workgroup and a partial row each (None: what the device wants, 128 at B = 32 on 256 CUs)
    for rep, nsp in enumerate((0, None, 5, B * T)):
        for flags in (torch.zeros(4 * B, device=dev), None):          # cross-time backward as a role / as a prologue
            r = ops.cap_cross_route_lin_bwd(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, dout, o_, pm, B, T, HS, HT, flags=flags, nsplit=nsp)
            assert r is not None
            dX2, dWp2, dbp2, dl2, ddyn2 = r
            assert nsp is None or dWp2.shape[0] == B * T + nsp
            assert torch.equal(dl2, dl1) and torch.equal(ddyn2, ddyn1)
            assert torch.equal(dX2, dX1), float((dX2 - dX1).abs().max())
            close(dWp2.sum(0), dWp1.sum(0).cpu(), tol=2e-6, what="route_lin dWp")
            close(dbp2.sum(0), dbp1.sum(0).cpu(), tol=2e-6, what="route_lin dbp")
            if nsp:         # the two halves' rows add up to the whole unit's row of the unsplit call
                k0 = B * T - nsp
                close((dWp2[k0::2] + dWp2[k0 + 1::2]), whole[1][k0:].cpu(), tol=2e-6, what="node halves dWp rows")
                assert torch.equal(dWp2[:k0], whole[1][:k0]) and torch.equal(dbp2[:k0], whole[2][:k0])
            elif nsp == 0:
                whole = r


@pytest.mark.parametrize("B,N,det", [(32, 170, 0), (3, 37, 0), (32, 170, 1)])
def test_cap_route_lin_bwd_carries_reduction_jobs(B, N, det):
    """gptst_cap_cross_route_lin_bwd_jobs (r05): a table of gradient-reduction jobs (gptst_pool_jobs kinds 1 / 2) rides in the routing backward's launch as
    role workgroups.  The launch's own outputs are bit-identical to the plain call; the pool gradients (owned elements) are bit-identical to gptst_pool_jobs,
    the embedding gradients (float atomics) agree to rounding.  det = 1 (gptst_set_deterministic): the same call runs its two launches instead."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(74)
    C, T, HS, HT = 64, 12, 10, 16
    s = rnd(B * T, HS, C, g=g).to(dev)
    dyn = (rnd(B, HT, T * HS, g=g) * 0.3).to(dev)
    tmpl = (torch.arange(1, T + 1) / 12.0).float().to(dev)
    c = torch.softmax(rnd(B * T, HS, N, g=g), 1).to(dev).contiguous()
    _, Ht, Rt = ops.cap_cross_fwd(s, dyn, tmpl, B, T, HS, HT)
    X = rnd(B, T, N, C, g=g).to(dev)
    Wp, bp = (rnd(C, C, g=g) * 0.1).to(dev), rnd(C, g=g).to(dev)
    dc1, dv = rnd(B * T, HS, N, g=g).to(dev), rnd(B * T, HS, C, g=g).to(dev)
    dout = rnd(B * T * N, C, g=g).to(dev)
    d = 10
    # reduction problems of the shapes a step queues: time-conditioned weights (B*T rows of C*C + C), node-conditioned weights in three row splits,
    # a scalar-column matrix (HS * N = 1700 / 370), and one-row "sum the partials" jobs
    te, ne = rnd(B * T, d, g=g).to(dev), rnd(N, d, g=g).to(dev)
    dWb = rnd(B * T, C * C + C, g=g).to(dev)
    dWn = rnd(3 * N, C * C + C, g=g).to(dev)
    dl = rnd(B * T, HS * N, g=g).to(dev)
    pool_w, pool_b, pool_l = rnd(d, C * C, g=g).to(dev), rnd(d, C, g=g).to(dev), rnd(d, HS * N, g=g).to(dev)
    ones = torch.ones(B * T, 1, device=dev)

    def table():
        pj = ops.PoolJobs()
        outs = [torch.zeros(d, C * C, device=dev), torch.zeros(d, C, device=dev), torch.zeros(d, C * C, device=dev), torch.zeros(d, HS * N, device=dev),
                torch.zeros(1, C * C, device=dev), torch.zeros(B * T, d, device=dev), torch.zeros(N, d, device=dev)]
        pj.bwd_pool(te, dWb[:, :C * C], outs[0])
        pj.bwd_pool(te, dWb[:, C * C:], outs[1])
        pj.bwd_pool(ne, dWn[:, :C * C], outs[2], nsplit=3)
        pj.bwd_pool(te, dl, outs[3])
        pj.bwd_pool(ones, dWb[:, :C * C], outs[4])
        pj.bwd_emb(dWb[:, :C * C], pool_w, outs[5])
        pj.bwd_emb(dWb[:, C * C:], pool_b, outs[5])
        pj.bwd_emb(dl, pool_l, outs[5])
        pj.bwd_emb(dWn[:, :C * C], pool_w, outs[6], nsplit=3)
        return pj, outs

    ref = ops.cap_cross_route_lin_bwd(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, dout, None, True, B, T, HS, HT, flags=torch.zeros(4 * B, device=dev))
    if ref is None:
        pytest.skip("the one-launch routing backward does not serve this shape")
    pj, ref_out = table()
    pj.launch()
    ops.set_deterministic(det)
    try:
        for rep in range(3):
            pj, outs = table()
            got = ops.cap_cross_route_lin_bwd(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, dout, None, True, B, T, HS, HT,
                                              flags=torch.zeros(4 * B, device=dev), jobs=pj)
            assert got is not None and not pj.jobs
            for a, b in zip(got, ref):
                assert torch.equal(a, b), rep
            for k, (a, b) in enumerate(zip(outs, ref_out)):
                if k < 5:
                    assert torch.equal(a, b), (rep, k)
                else:
                    close(a, b.cpu(), tol=2e-6, what="carried embedding gradient")
    finally:
        ops.set_deterministic(0)
    assert _handoff_timeouts() == 0


@pytest.mark.parametrize("B,N", [(32, 170), (3, 37), (9, 16)])
def test_hypertem_bwd_pair_equals_two_layer_calls(B, N):
    """gptst_hypertem_bwd_pair (two adjacent hyperTem layers' backward on the slab, the lower layer's weight-gradient role fed by write-through
    stores + a per-sample counter) == two gptst_hypertem_bwd_wgrad calls in the dPre chain, bit for bit, call after call."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    T, C = 12, 64
    x0 = rnd(B, T, N, C, g=g).to(dev)
    G0, G1 = (rnd(N, T, T, g=g) * 0.2).to(dev), (rnd(N, T, T, g=g) * 0.2).to(dev)
    W0, W1 = (rnd(B * T, C, C, g=g) * 0.1).to(dev), (rnd(B * T, C, C, g=g) * 0.1).to(dev)
    b0, b1 = rnd(B * T, C, g=g).to(dev), rnd(B * T, C, g=g).to(dev)
    R0, x1 = ops.hypertem_fwd(x0, G0, W0, b0)
    R1, x2 = ops.hypertem_fwd(x1, G1, W1, b1)
    dpre1 = rnd(B, T, N, C, g=g).to(dev)
    dmid, dWb1, ns, dG1 = ops.hypertem_bwd_wgrad(dpre1, None, x1, G1, W1, R1, premul=True)
    dx0, dWb0, _, dG0 = ops.hypertem_bwd_wgrad(dmid, None, x0, G0, W0, R0, premul=True)
    for rep in range(3):
        pG1, pG0 = torch.empty_like(dG1), torch.empty_like(dG0)
        r = ops.hypertem_bwd_pair(dpre1, x1, G1, W1, R1, x0, G0, W0, R0, pG1, pG0, torch.zeros(B, device=dev))
        assert r is not None
        pmid, px0, pW1, pW0, pns = r
        assert pns == ns
        for a, b_, nm in ((pmid, dmid, "dXmid"), (px0, dx0, "dX0"), (pW1, dWb1, "dWb1"), (pW0, dWb0, "dWb0"), (pG1, dG1, "dG1"), (pG0, dG0, "dG0")):
            assert torch.equal(a, b_), "pair form differs in %s (rep %d)" % (nm, rep)


def test_temporal_graph_job_falls_back_beyond_its_lds_scratch():
    """the kind-3 job of gptst_pool_jobs (hyperTem's G_n = A_n^T A_n from node embedding and hyperedge pool) holds pool + rows in
    4160 LDS floats — 20 hyperedges at embed_dim 16.  Beyond that PoolJobs.gram builds the graph with gptst_gram_fwd behind the launch: same
    numbers as the job form where both serve, and the large shape equals the torch restatement."""
    from gptst_amd import ops, _C
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    for N, d, Hm in ((170, 16, 16), (170, 16, 20), (170, 16, 24), (37, 8, 48)):
        ne, pool = rnd(N, d, g=g).to(dev), (rnd(d, Hm * 12, g=g) * 0.3).to(dev)
        fits = _C.lib().value("gptst_pool_jobs_gram_rows", d, Hm * 12) > 0
        assert fits == ((d + 1) * Hm * 12 <= 4160)
        jobs = ops.PoolJobs()
        A = jobs.fwd(ne, pool)
        G = torch.empty(N, 12, 12, device=dev)
        jobs.gram(ne, pool, out=G, A=A)
        assert (len(jobs.post) == 0) == fits
        jobs.launch()
        Ar = (ne.cpu() @ pool.cpu()).view(N, Hm, 12)
        close(G, torch.einsum("nht,nhu->ntu", Ar, Ar), what="temporal graph (job or fallback)")
        assert torch.equal(G, ops.gram_fwd(A.view(N, Hm, 12)))


@pytest.mark.parametrize("shape,base", [((32, 12, 170), 1), ((3, 12, 37), 2), ((1, 5, 7), 1)])
def test_fusion_gate_equals_the_torch_modules(shape, base):
    """gptst_fusion_gate_fwd / _bwd (the downstream front end's lin_test + Fusion gate, reference model/Model.py:5-18,:106) against the torch
    modules on the same parameters: fused embedding and every parameter gradient (fp32, 1e-4 of the tensor scale); inference (no parameter
    requires grad) takes the same launch without keeping z."""
    from gptst_amd.enhance import Fusion
    from gptst_amd.fusion import fusion_gate
    dev = _dev()
    torch.manual_seed(5)
    C = 64
    fus, lin = Fusion(C), torch.nn.Linear(base, C)
    F = torch.randn(*shape, C) * 0.7
    src = torch.randn(*shape, base + 2)
    go = torch.randn(*shape, C)
    ref = fus(F, lin(src[..., :base]))
    (ref * go).sum().backward()
    want = {n: p.grad.clone() for n, p in list(fus.named_parameters()) + [("lin." + k, v) for k, v in lin.named_parameters()]}
    fus_d, lin_d = Fusion(C).to(dev), torch.nn.Linear(base, C).to(dev)
    fus_d.load_state_dict(fus.state_dict()); lin_d.load_state_dict(lin.state_dict())
    out = fusion_gate(F.to(dev), src.to(dev), fus_d, lin_d, base)
    close(out, ref.detach(), what="fusion gate out")
    (out * go.to(dev)).sum().backward()
    got = {n: p.grad for n, p in list(fus_d.named_parameters()) + [("lin." + k, v) for k, v in lin_d.named_parameters()]}
    for n in want:
        close(got[n], want[n], what="fusion gate d" + n)
    for p_ in list(fus_d.parameters()) + list(lin_d.parameters()):
        p_.requires_grad_(False)
    with torch.no_grad():
        close(fusion_gate(F.to(dev), src.to(dev), fus_d, lin_d, base), ref.detach(), what="fusion gate out (inference)")


def test_step_begin_draws_philox_noise():
    """gptst_step_begin fills the step's mask noise with Philox4x32-10 uniforms keyed by device words (seed, step): exact against a Python
    restatement of the published algorithm, in [0,1), uniform, and a different stream per step."""
    from gptst_amd import ops
    dev = _dev()
    src = torch.zeros(2, 12, 5, 3, device=dev)
    z0 = torch.ones(64, device=dev)
    n = 65280 * 2 + 3
    noise = torch.full((n,), -1.0, device=dev)
    rng = torch.tensor([1234567, 42], dtype=torch.int32, device=dev)
    ops.step_begin(z0, None, src, 1, noise=noise, rng=rng)
    u = noise.cpu()
    assert float(z0.abs().max()) == 0.0
    for i in (0, 1, 17, n // 4 - 1, n // 4):                       # n // 4: the ragged tail (3 values)
        ref = _philox_ref(i, 42, 1234567)
        got = u[4 * i:4 * i + 4].tolist()
        assert got == [float(torch.tensor(v, dtype=torch.float32)) for v in ref[:len(got)]], (i, got, ref)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 5e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    noise2 = torch.empty_like(noise)
    ops.step_begin(z0, None, src, 1, noise=noise2, rng=torch.tensor([1234567, 43], dtype=torch.int32, device=dev))
    assert float((noise2.cpu() == u).float().mean()) < 1e-3


@pytest.mark.parametrize("W", [2, 8, 32, 64, 100])   # 32 / 64 / 100: 261 120 / 522 240 / 816 000 cells = the global batch of four / eight bench-shape ranks / eight NYC_TAXI-shape
                                                      # ranks (two / four / eight cells per thread of the cooperative launch, r06)
def test_mask_generation_over_a_global_batch(W):
    """Data parallel with global masks: every rank selects over world x B*T*N cells — more selection workgroups than the single-rank 64
    (one per 1024 cells, up to 512); bit-exact against the oracle for both phases."""
    from gptst_amd import ops, synth
    dev = _dev()
    B, T, N, HS = 4 * W, 12, 170, 10
    M = B * T * N
    noise = synth.make_noise(M, 3)
    assert torch.equal(ops.mask_random(noise.to(dev), int(M * 0.25)).cpu().to(torch.int64), O.random_mask(noise, 0.25))
    g = torch.Generator().manual_seed(W)
    label_ref = torch.randint(0, HS, (M,), generator=g)
    list_c = synth.class_order(HS, 5)
    na, nr = synth.make_noise(M, 11), synth.make_noise(M, 12)
    total = int(M * 0.25)
    ada = total // 2
    m_ada_r, m_rnd_r, fin_r = O.adaptive_mask(label_ref.view(B, T, N), list_c, na, nr, ada, total - ada, "all")
    m_ada, m_rnd, mask = ops.mask_adaptive(label_ref.to(torch.int32).to(dev), None, torch.tensor(list_c, dtype=torch.int32, device=dev),
                                           torch.tensor([ada, total - ada], dtype=torch.int32, device=dev), na.to(dev), nr.to(dev), 1, 1)
    assert torch.equal(m_ada.cpu().long(), m_ada_r) and torch.equal(m_rnd.cpu().long(), m_rnd_r)
    assert torch.equal(mask.cpu().long(), fin_r.view(-1))
    # r04: what the data-parallel step calls — the two-digit select on the 24-bit noise lattice at the global size
    assert torch.equal(ops.mask_random(noise.to(dev), int(M * 0.25), u24=True).cpu().to(torch.int64), O.random_mask(noise, 0.25))
    m_ada, m_rnd, mask = ops.mask_adaptive(label_ref.to(torch.int32).to(dev), None, torch.tensor(list_c, dtype=torch.int32, device=dev),
                                           torch.tensor([ada, total - ada], dtype=torch.int32, device=dev), na.to(dev), nr.to(dev), 1, 1, u24=True)
    assert torch.equal(m_ada.cpu().long(), m_ada_r) and torch.equal(m_rnd.cpu().long(), m_rnd_r)
    assert torch.equal(mask.cpu().long(), fin_r.view(-1))


@pytest.mark.parametrize("B,N,HS,R", [(32, 170, 10, 2), (2, 207, 10, 2), (1, 256, 16, 3), (2, 20, 5, 0), (2, 20, 5, 1), (1, 100, 3, 4), (1, 250, 7, 2)])
def test_cap_route_fwd4_matches_second_generation(B, N, HS, R):
    """cap_route_fwd4_kernel (8 waves, wave-local routing passes over LDS-resident tiles, two workgroups per CU; one fold per routing iteration)
    against the LDS-resident second generation (gptst_tune(20, 1), the kernel of the shapes the fourth does not serve) on the same inputs: same
    algebra, different summation order over the nodes (tile partials folded over the waves) — soft assignment c and cluster aggregate s to 2e-6 of
    their scale (routing is detached: no gradient flows through it).  (r05: the one-wave-per-tile third variant left the library.)"""
    from gptst_amd import ops, _C
    dev = _dev()
    g = torch.Generator().manual_seed(71 + N)
    C, T = 64, 12
    X = rnd(B, T, N, C, g=g, scale=0.5).to(dev)
    Wp, bp = (rnd(C, C, g=g) * 0.15).to(dev), (rnd(C, g=g) * 0.3).to(dev)
    dadj = rnd(B * T, HS * N, g=g).to(dev)
    lib = _C.lib()
    try:
        lib.call("gptst_tune", 20, 1)
        c2, s2 = ops.cap_route_fwd(X, Wp, bp, dadj, HS, R)
        lib.call("gptst_tune", 20, 0)
        c3, s3 = ops.cap_route_fwd(X, Wp, bp, dadj, HS, R)
    finally:
        lib.call("gptst_tune", 20, 0)
    assert not torch.equal(c2, c3) or N <= 16, "both calls ran the same kernel?"
    close(c3, c2.cpu(), tol=2e-6, what="route3 c")
    close(s3, s2.cpu(), tol=2e-6, what="route3 s")


@pytest.mark.parametrize("B,N,nstage", [(32, 170, 2), (2, 37, 3), (5, 16, 1), (1, 250, 2)])
def test_hypertem_chain_fwd_equals_layer_calls(B, N, nstage):
    """gptst_hypertem_chain_fwd (up to three hyperTem layers on the LDS slab, one launch) against the per-layer entry point: the same arithmetic
    in the same order — bit-identical."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(81 + N)
    C, T = 64, 12
    x = rnd(B, T, N, C, g=g, scale=0.7).to(dev)
    stages = [((rnd(N, T, T, g=g) * 0.2).to(dev), (rnd(B * T, C, C, g=g) * 0.1).to(dev), (rnd(B * T, C, g=g) * 0.3).to(dev)) for _ in range(nstage)]
    res = ops.hypertem_chain_fwd(x, stages)
    for k, ((G, Wbt, bbt), (R, o)) in enumerate(zip(stages, res)):
        R1, o1 = ops.hypertem_fwd(x, G, Wbt, bbt)
        assert torch.equal(R, R1), "R of chained layer %d" % k
        assert torch.equal(o, o1), "out of chained layer %d" % k
        x = o1


@pytest.mark.parametrize("B,N,C,masked", [(32, 170, 64, True), (2, 20, 64, True), (3, 37, 64, False), (1, 300, 128, True)])
def test_encin_ht1_low_rank_pair_equals_lin_in_plus_hypertem(B, N, C, masked):
    """encin.hip: input projection (base = 1) + the encoder's first hyperTem layer on the rank-2 structure of the input, forward and backward,
    against the generic kernels (gptst_lin_in + gptst_hypertem_fwd; their backward in chain form + gptst_rowouter for the projection's
    gradient): same mathematics, sums associated differently."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(91 + N)
    T = 12
    src = rnd(B, T, N, 3, g=g).to(dev)
    mask = (torch.rand(B * T * N, generator=g) > 0.3).float().to(dev) if masked else None
    fill = -1.5753
    w, bi = rnd(C, 1, g=g).to(dev), (rnd(C, g=g) * 0.5).to(dev)
    G = (rnd(N, T, T, g=g) * 0.3).to(dev)
    Wbt, bbt = (rnd(B * T, C, C, g=g) * 0.1).to(dev), (rnd(B * T, C, g=g) * 0.3).to(dev)
    out, ab, wv = ops.encin_ht1_fwd(src, 1, mask, fill if masked else 0.0, w, bi, G, Wbt, bbt)
    x0 = ops.lin_in(src, 3, 1, w, bi, C, mask=mask, fill=fill)
    if C == 64:
        R, ref = ops.hypertem_fwd(x0.view(B, T, N, C), G, Wbt, bbt)
    else:
        from gptst_amd.ops import MODE_TIME, EPI_RES_LRELU
        R = ops.tmix(x0.view(B, T, N, C), G)
        ref = ops.apply(R.view(-1, C), Wbt, MODE_TIME, B * T, N, bias=bbt, resid=x0, epi=EPI_RES_LRELU).view(B, T, N, C)
    close(out, ref.cpu(), tol=3e-6, what="encin out")
    # backward, chain form: dPre = dOut * lrelu'(out)
    dO = rnd(B, T, N, C, g=g).to(dev)
    dPre = (dO * torch.where(ref > 0, torch.ones_like(ref), torch.full_like(ref, 0.01))).contiguous()
    dWb, dG, dinp = ops.encin_ht1_bwd(dPre, src, mask, fill if masked else 0.0, w, bi, Wbt, ab, wv)
    x0v, Rv, dP = x0.view(B * T, N, C).double(), R.reshape(B * T, N, C).double(), dPre.view(B * T, N, C).double()
    close(dWb[:, :C * C].view(B * T, C, C), torch.einsum("gni,gno->gio", Rv, dP).cpu(), tol=5e-6, what="encin dW_bt")
    close(dWb[:, C * C:], dP.sum(1).cpu(), tol=5e-6, what="encin db_bt")
    dR = torch.einsum("gno,gio->gni", dP, Wbt.double()).view(B, T, N, C)
    close(dG, torch.einsum("btnc,bunc->bntu", dR, x0v.view(B, T, N, C)).cpu(), tol=5e-6, what="encin dG")
    dX0 = dP.view(B, T, N, C) + torch.einsum("ntu,btnc->bunc", G.double(), dR)
    m = (torch.where(mask.view(B, T, N) != 0, src[..., 0], torch.full_like(src[..., 0], fill)) if masked else src[..., 0]).double()
    close(dinp[:, :C].sum(0), torch.einsum("btn,btnc->c", m, dX0).cpu(), tol=5e-6, what="encin d weight")
    close(dinp[:, C:].sum(0), dX0.sum((0, 1, 2)).cpu(), tol=5e-6, what="encin d bias")


@pytest.mark.parametrize("B,N,C", [(32, 170, 64), (2, 20, 64), (1, 45, 128)])
def test_guide_in_low_rank_pair_equals_lin_in_plus_node_layer(B, N, C):
    """guidein.hip: MLP_RL's input projection (base = 1) + node-conditioned layer as an elementwise pass, and their backward from two vectors
    per node, against gptst_lin_in + gptst_apply(MODE_NODE) and fp64 einsums."""
    from gptst_amd import ops
    from gptst_amd.ops import MODE_NODE, EPI_LRELU
    dev = _dev()
    g = torch.Generator().manual_seed(93 + N)
    T = 12
    src = rnd(B, T, N, 3, g=g).to(dev)
    w1, b1 = rnd(C, 1, g=g).to(dev), (rnd(C, g=g) * 0.5).to(dev)
    Wn, bn = (rnd(N, C, C, g=g) * 0.1).to(dev), (rnd(N, C, g=g) * 0.3).to(dev)
    h1 = ops.guide_in_fwd(src, w1, b1, Wn, bn)
    h0 = ops.lin_in(src, 3, 1, w1, b1, C)
    ref = ops.apply(h0, Wn, MODE_NODE, B * T, N, bias=bn, epi=EPI_LRELU)
    close(h1, ref.cpu(), tol=3e-6, what="guide_in h1")
    dPre = rnd(B * T * N, C, g=g).to(dev)
    dWb, dinp = ops.guide_in_bwd(dPre, src, w1, b1, Wn)
    h0v, dP, s = h0.view(B * T, N, C).double(), dPre.view(B * T, N, C).double(), src[..., 0].reshape(B * T, N).double()
    close(dWb[:, :C * C].view(N, C, C), torch.einsum("rni,rno->nio", h0v, dP).cpu(), tol=5e-6, what="guide_in dW_n")
    close(dWb[:, C * C:], dP.sum(0).cpu(), tol=5e-6, what="guide_in db_n")
    dh0 = torch.einsum("rno,nio->rni", dP, Wn.double())
    close(dinp[:, :C].sum(0), torch.einsum("rn,rni->i", s, dh0).cpu(), tol=5e-6, what="guide_in d ln1.weight")
    close(dinp[:, C:].sum(0), dh0.sum((0, 1)).cpu(), tol=5e-6, what="guide_in d ln1.bias")


@pytest.mark.parametrize("B,N,HS", [(32, 170, 10), (2, 50, 5), (3, 37, 16), (2, 207, 10)])
def test_guide_head_fwd_equals_the_three_launches(B, N, HS):
    """r06: gptst_guide_head_fwd (node vectors + one (b,t)-grouped pass) == gptst_guide_in_fwd + gptst_apply(TIME, LReLU) + gptst_rowdot(softmax,
    label): the same fmaf / MFMA chains in the same order -> h1, h2, prob and the labels are bit-identical."""
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(81)
    C, T = 64, 12
    src = rnd(B, T, N, 3, g=g).to(dev)
    w1, b1 = rnd(C, 1, g=g).to(dev), rnd(C, g=g).to(dev)
    Wn, bn = (rnd(N, C, C, g=g) * 0.2).to(dev), rnd(N, C, g=g).to(dev)
    Wbt, bbt = (rnd(B * T, C, C, g=g) * 0.2).to(dev), rnd(B * T, C, g=g).to(dev)
    W3, b3 = (rnd(HS, C, g=g) * 0.3).to(dev), rnd(HS, g=g).to(dev)
    h1r = ops.guide_in_fwd(src, w1, b1, Wn, bn)
    h2r = ops.apply(h1r, Wbt, ops.MODE_TIME, B * T, N, bias=bbt, epi=ops.EPI_LRELU)
    pr, lr = ops.rowdot(h2r, W3, b3, softmax=True, want_label=True)
    for _ in range(2):
        r = ops.guide_head_fwd(src, w1, b1, Wn, bn, Wbt, bbt, W3, b3)
        assert r is not None
        h1, h2, prob, label = r
        assert torch.equal(h1, h1r), float((h1 - h1r).abs().max())
        assert torch.equal(h2, h2r), float((h2 - h2r).abs().max())
        assert torch.equal(prob, pr) and torch.equal(label, lr)
