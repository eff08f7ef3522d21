"""GPU parity tests (run on the MI355X box: pytest -m gpu): HIP kernels through the C ABI vs the pinned CPU oracle.
Tolerance: fp32, 1e-4 relative to the tensor scale (north-star), index/mask work bit-exact."""
import pytest
import torch

from oracle import gptst_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def close(a, b, tol=1e-4, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = b.abs().max().clamp_min(1e-6)
    err = float((a - b).abs().max() / scale)
    assert err < tol, "%s: max err / scale = %.3e (scale %.3e)" % (what, err, float(scale))
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
@pytest.mark.parametrize("BT,N", [(24, 20), (36, 170), (5, 33)])
def test_apply_and_wgrad(mode, BT, N):
    """MFMA contractions vs fp32 matmul, asymmetric random weights (catches transposed fragments)."""
    from gptst_amd import ops
    dev = _dev()
    C = 64
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


def test_tmix_and_graph():
    from gptst_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, T, N, C, Hm = 3, 12, 21, 64, 8
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
    (ref * go).sum().backward()
    gpu = [t.to(dev).requires_grad_() for t in ts]
    out = layers.hypertem(*gpu)
    (out * go.to(dev)).sum().backward()
    close(out, ref, what="hypertem out")
    for nm, a, b in zip(["x", "node_emb", "time_eb", "adj", "wpool", "bpool"], gpu, cpu):
        close(a.grad, b.grad, tol=2e-4, what="hypertem d" + nm)


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
def test_cap_layer(B, N, d, ds, HS, HT, R):
    from gptst_amd import layers
    dev = _dev()
    C, T = 64, 12
    ts, go = _cap_case(B, N, C, d, ds, HS, HT, 21)
    tmpl = torch.linspace(1, T, steps=T) / 12.0
    cpu = [t.clone().requires_grad_() for t in ts]
    sd = {"c.t_adj": cpu[4], "c.adj": cpu[5], "c.weights_spa": cpu[6], "c.bias_spa": cpu[7], "c.ln_p.weight": cpu[8],
          "c.ln_p.bias": cpu[9], "c.mask_template": tmpl}
    ref, cref, dynref, aux = O.cap(sd, "c.", cpu[0], cpu[1], cpu[2], cpu[3], R, materialize_5d=(N <= 64), return_aux=True)
    (ref * go).sum().backward()
    gpu = [t.to(dev).requires_grad_() for t in ts]
    out, c, dyn = layers.cap(*gpu, tmpl.to(dev), R)
    (out * go.to(dev)).sum().backward()
    close(c, cref.squeeze(-1), what="cap c")
    close(dyn, dynref, what="cap dyn")
    close(out, ref, what="cap out")
    names = ["x", "node_emb", "time_eb_spg", "teb", "t_adj", "adj", "wspa", "bspa", "lnp_w", "lnp_b"]
    for nm, a, b in zip(names, gpu, cpu):
        close(a.grad, b.grad, tol=3e-4, what="cap d" + nm)
