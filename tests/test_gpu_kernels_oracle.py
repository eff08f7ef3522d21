"""Oracle-level checks of the fused launches the default step runs since round 4 (their kernel tests compared HIP with HIP):
the hyperTem chain forward + pair backward, the encoder's low-rank first layer (`encin`) and the guide's (`guidein`), each against the pinned
oracle's layer functions (oracle/gptst_oracle.py: `hypertem`, `_lin`, the MLP_RL lines) run on the CPU with autograd — forward outputs and EVERY
parameter gradient.  The kernels emit generated-parameter gradients (dW_bt, db_bt, dG, dW_n ...); they are carried to the parameters the oracle
differentiates (pools, adjacency, embeddings) with fp64 einsums here, so that a kernel error shows up against an oracle number with nothing but exact
linear maps in between.  Shapes include N = 50 / d = 5-style small cases (where round 4's gradient outlier lived) and the bench shape's N = 170."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_kernels import _hypertem_case, close, rnd  # noqa: E402
from oracle import gptst_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SLOPE = O.LRELU_SLOPE
T = 12


def _gen(ne, te, adj, wp, bp, B, N, C):
    """generated parameters of one hyperTem layer from its pools, on the device, through the product's own generation launch"""
    from gptst_amd import ops
    d, Hm = adj.shape[0], adj.shape[1]
    jobs = ops.PoolJobs()
    A = jobs.fwd(ne, adj.view(d, Hm * T))
    Wbt = jobs.fwd(te.view(B * T, d), wp.view(d, C * C))
    bbt = jobs.fwd(te.view(B * T, d), bp)
    G = torch.empty(N, T, T, device=ne.device)
    jobs.gram(ne, adj.view(d, Hm * T), out=G, A=A)
    jobs.launch()
    return A.view(N, Hm, T), G, Wbt.view(B * T, C, C), bbt


def _to_params(dWb, dG, A, ne, te, adj, wp, bp, B, N, C):
    """[dW_bt | db_bt] rows (any number of row splits) and graph-gradient partials of one layer -> gradients of (ne, te, adj, wp, bp), fp64"""
    d, Hm = adj.shape[0], adj.shape[1]
    BT = B * T
    dWb = dWb.double().view(-1, BT, C * C + C).sum(0).cpu()
    dW, db = dWb[:, :C * C].view(BT, C, C), dWb[:, C * C:]
    dG = dG.double().view(-1, N, T, T).sum(0).cpu()
    te2, A = te.double().cpu().view(BT, d), A.double().cpu()
    g_wp = torch.einsum("gd,gio->dio", te2, dW)
    g_bp = te2.t() @ db
    g_te = (torch.einsum("gio,dio->gd", dW, wp.double().cpu()) + db @ bp.double().cpu().t()).view(B, T, d)
    dA = torch.einsum("ntu,nhu->nht", dG + dG.transpose(1, 2), A)          # G_n = A_n^T A_n
    g_adj = torch.einsum("nk,nht->kht", ne.double().cpu(), dA)
    g_ne = torch.einsum("nht,kht->nk", dA, adj.double().cpu())
    return g_ne, g_te, g_adj, g_wp, g_bp


@pytest.mark.parametrize("B,N,d,Hm", [(2, 50, 5, 8), (3, 170, 16, 8), (1, 33, 4, 5), (32, 170, 10, 8)])
def test_hypertem_chain_fwd_and_pair_bwd_vs_oracle(B, N, d, Hm):
    """gptst_hypertem_chain_fwd (two layers on the slab) and gptst_hypertem_bwd_pair (their backward in one launch, dPre chain) against
    oracle.hypertem o oracle.hypertem (GPTST.py:154-163 twice, as :267-269) with autograd."""
    from gptst_amd import ops
    C = 64
    (z, ne, te, adj0, wp0, bp0), go = _hypertem_case(B, N, C, d, Hm, 31)
    (_, _, _, adj1, wp1, bp1), _ = _hypertem_case(B, N, C, d, Hm, 32)
    cpu = [t.clone().requires_grad_() for t in (z, ne, te, adj0, wp0, bp0, adj1, wp1, bp1)]
    x = F.leaky_relu(cpu[0], SLOPE)                       # the chain's input is the output of a LeakyReLU layer (the dPre chain hands dX * lrelu'(X) down)
    h1 = O.hypertem({"h.adj": cpu[3], "h.weights_pool": cpu[4], "h.bias_pool": cpu[5]}, "h.", x, cpu[1], cpu[2])
    h2 = O.hypertem({"h.adj": cpu[6], "h.weights_pool": cpu[7], "h.bias_pool": cpu[8]}, "h.", h1, cpu[1], cpu[2])
    go = go * (h2.detach().abs() > 1e-5) * (h1.detach().abs() > 1e-5) * (x.detach().abs() > 1e-5)     # (a pre-activation within fp32 noise of 0: not a parity question)
    (h2 * go).sum().backward()

    dv = lambda t: t.to(DEV).contiguous()      # noqa: E731
    xd, ned, ted = dv(x.detach()), dv(ne), dv(te)
    L0, L1 = [dv(t) for t in (adj0, wp0, bp0)], [dv(t) for t in (adj1, wp1, bp1)]
    A0, G0, W0, b0 = _gen(ned, ted, *L0, B, N, C)
    A1, G1, W1, b1 = _gen(ned, ted, *L1, B, N, C)
    (R0, o0), (R1, o1) = ops.hypertem_chain_fwd(xd, [(G0, W0, b0), (G1, W1, b1)])
    close(o0, h1, what="chain fwd layer 1 vs oracle")
    close(o1, h2, what="chain fwd layer 2 vs oracle")
    dpre1 = (dv(go) * torch.where(o1 > 0, torch.ones_like(o1), torch.full_like(o1, SLOPE))).contiguous()
    dG1, dG0 = torch.empty(B, N, T, T, device=DEV), torch.empty(B, N, T, T, device=DEV)
    r = ops.hypertem_bwd_pair(dpre1, o0, G1, W1, R1, xd, G0, W0, R0, dG1, dG0, torch.zeros(B, device=DEV))
    if r is None:                                           # shapes the pair launch refuses run the two layer calls (what the engine does)
        dmid, dWb1, _, dG1 = ops.hypertem_bwd_wgrad(dpre1, None, o0, G1, W1, R1, premul=True)
        dx0, dWb0, _, dG0 = ops.hypertem_bwd_wgrad(dmid, None, xd, G0, W0, R0, premul=True)
    else:
        dmid, dx0, dWb1, dWb0, _ = r
    close(dx0, cpu[0].grad, what="pair bwd dX (x lrelu'(X)) vs oracle")
    g1 = _to_params(dWb1, dG1, A1, ned, ted, *L1, B, N, C)
    g0 = _to_params(dWb0, dG0, A0, ned, ted, *L0, B, N, C)
    close(g0[0] + g1[0], cpu[1].grad, what="pair bwd d node_emb vs oracle")
    close(g0[1] + g1[1], cpu[2].grad, what="pair bwd d time_eb vs oracle")
    for k, (g_, nm) in enumerate(zip(g0[2:], ("adj", "weights_pool", "bias_pool"))):
        close(g_, cpu[3 + k].grad, what="pair bwd lower layer d%s vs oracle" % nm)
    for k, (g_, nm) in enumerate(zip(g1[2:], ("adj", "weights_pool", "bias_pool"))):
        close(g_, cpu[6 + k].grad, what="pair bwd upper layer d%s vs oracle" % nm)


@pytest.mark.parametrize("B,N,d,Hm,masked", [(2, 50, 5, 8, True), (3, 170, 16, 8, True), (1, 33, 4, 5, False)])
def test_encin_low_rank_first_layer_vs_oracle(B, N, d, Hm, masked):
    """encin.hip (input projection + the encoder's hyperTem1 on the rank-2 structure of the masked input, base = 1) against the oracle's
    lines GPTST.py:416-418 (mask, mask token, dim_in_flow) + oracle.hypertem, forward and every gradient."""
    from gptst_amd import ops
    C, fill = 64, -1.5753
    (_, ne, te, adj, wp, bp), go = _hypertem_case(B, N, C, d, Hm, 41)
    g = torch.Generator().manual_seed(43 + N)
    src = rnd(B, T, N, 3, g=g)
    keep = (torch.rand(B, T, N, 1, generator=g) > 0.3).float() if masked else torch.ones(B, T, N, 1)
    w, bi = rnd(C, 1, g=g), rnd(C, g=g) * 0.5
    cpu = [t.clone().requires_grad_() for t in (w, bi, ne, te, adj, wp, bp)]
    msrc = keep * src[..., 0:1]                                                             # :416
    msrc = torch.where(keep == 0, torch.full_like(msrc, fill), msrc)                        # :417
    x0 = O._lin({"l.weight": cpu[0], "l.bias": cpu[1]}, "l", msrc)                          # :418
    ref = O.hypertem({"h.adj": cpu[4], "h.weights_pool": cpu[5], "h.bias_pool": cpu[6]}, "h.", x0, cpu[2], cpu[3])
    go = go * (ref.detach().abs() > 1e-5)
    (ref * go).sum().backward()

    dv = lambda t: t.to(DEV).contiguous()      # noqa: E731
    srcd, ned, ted, wd, bid = dv(src), dv(ne), dv(te), dv(w), dv(bi)
    L = [dv(t) for t in (adj, wp, bp)]
    A, G, Wbt, bbt = _gen(ned, ted, *L, B, N, C)
    maskd = dv(keep.view(-1)) if masked else None
    out, ab, wv = ops.encin_ht1_fwd(srcd, 1, maskd, fill if masked else 0.0, wd, bid, G, Wbt, bbt)
    close(out, ref, what="encin fwd vs oracle")
    dpre = (dv(go) * torch.where(out > 0, torch.ones_like(out), torch.full_like(out, SLOPE))).contiguous()
    dWb, dG, dinp = ops.encin_ht1_bwd(dpre, srcd, maskd, fill if masked else 0.0, wd, bid, Wbt, ab, wv)
    g_ne, g_te, g_adj, g_wp, g_bp = _to_params(dWb, dG, A, ned, ted, *L, B, N, C)
    close(dinp[:, :C].double().sum(0).view(C, 1), cpu[0].grad, what="encin d dim_in_flow.weight vs oracle")
    close(dinp[:, C:].double().sum(0), cpu[1].grad, what="encin d dim_in_flow.bias vs oracle")
    close(g_ne, cpu[2].grad, what="encin d node_emb vs oracle")
    close(g_te, cpu[3].grad, what="encin d time_eb vs oracle")
    close(g_adj, cpu[4].grad, what="encin d adj vs oracle")
    close(g_wp, cpu[5].grad, what="encin d weights_pool vs oracle")
    close(g_bp, cpu[6].grad, what="encin d bias_pool vs oracle")


@pytest.mark.parametrize("B,N,d", [(2, 50, 5), (3, 170, 16), (1, 33, 4)])
def test_guide_in_low_rank_first_layers_vs_oracle(B, N, d):
    """guidein.hip (MLP_RL.ln1 + the node-conditioned layer as one elementwise pass, base = 1) against the oracle's lines GPTST.py:22-27
    (oracle.mlp_rl's first two stages), forward and every gradient."""
    from gptst_amd import ops
    C = 64
    g = torch.Generator().manual_seed(51 + N)
    src = rnd(B, T, N, 3, g=g)
    w1, b1, nb = rnd(C, 1, g=g), rnd(C, g=g) * 0.5, rnd(N, d, g=g)
    wps, bps = rnd(d, C, C, g=g, scale=0.1), rnd(d, C, g=g, scale=0.3)
    go = rnd(B, T, N, C, g=g)
    cpu = [t.clone().requires_grad_() for t in (w1, b1, nb, wps, bps)]
    h = O._lin({"l.weight": cpu[0], "l.bias": cpu[1]}, "l", src[..., 0:1])                 # :22
    wn = torch.einsum("nd,dio->nio", cpu[2], cpu[3])                                        # :24
    bn = torch.matmul(cpu[2], cpu[4])                                                       # :25
    ref = F.leaky_relu(torch.einsum("btni,nio->btno", h, wn) + bn, SLOPE)                   # :26-27
    go = go * (ref.detach().abs() > 1e-5)
    (ref * go).sum().backward()

    dv = lambda t: t.to(DEV).contiguous()      # noqa: E731
    srcd, w1d, b1d, nbd, wpsd, bpsd = [dv(t) for t in (src, w1, b1, nb, wps, bps)]
    jobs = ops.PoolJobs()
    Wn, bnd = jobs.fwd(nbd, wpsd.view(d, C * C)), jobs.fwd(nbd, bpsd)
    jobs.launch()
    h1 = ops.guide_in_fwd(srcd, w1d, b1d, Wn.view(N, C, C), bnd)
    close(h1.view(B, T, N, C), ref, what="guide_in fwd vs oracle")
    h1v = h1.view(B, T, N, C)
    dpre = (dv(go) * torch.where(h1v > 0, torch.ones_like(h1v), torch.full_like(h1v, SLOPE))).contiguous().view(-1, C)
    dWb, dinp = ops.guide_in_bwd(dpre, srcd, w1d, b1d, Wn.view(N, C, C))
    dWb = dWb.double().cpu()
    dW, db = dWb[:, :C * C].view(N, C, C), dWb[:, C * C:]
    nb64 = nb.double()
    close(torch.einsum("nd,nio->dio", nb64, dW), cpu[3].grad, what="guide_in d weights_pool_spa vs oracle")
    close(nb64.t() @ db, cpu[4].grad, what="guide_in d bias_pool_spa vs oracle")
    close(torch.einsum("nio,dio->nd", dW, wps.double()) + db @ bps.double().t(), cpu[2].grad, what="guide_in d neb4mask vs oracle")
    close(dinp[:, :C].double().sum(0).view(C, 1), cpu[0].grad, what="guide_in d ln1.weight vs oracle")
    close(dinp[:, C:].double().sum(0), cpu[1].grad, what="guide_in d ln1.bias vs oracle")
