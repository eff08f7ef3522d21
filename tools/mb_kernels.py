#!/usr/bin/env python
"""Micro-benchmark of the hot kernels at the bench shape (B=32,T=12,N=170,C=64,HS=10): each C-ABI call is captured 20x back to back
in a hipGraph and replayed; prints us per launch.  usage (GPU box): python tools/mb_kernels.py [name-filter]"""
import sys
sys.path.insert(0, '.')
import torch
from gptst_amd import ops
from gptst_amd.ops import MODE_TIME, MODE_NODE, MODE_SHARED, PRO_DPRE, EPI_RES_LRELU, EPI_ADD_DPRE

dev = 'cuda:0'
B, T, N, C, HS, HT, R = 32, 12, 170, 64, 10, 16, 2
BT = B * T
torch.manual_seed(0)
f = lambda *s: torch.randn(*s, device=dev)
X, dO = f(B, T, N, C), f(B, T, N, C)
G = f(N, 12, 12) * 0.1
Wbt, bbt = f(BT, C, C) * 0.1, f(BT, C)
Wn, bn = f(N, C, C) * 0.1, f(N, C)
Wp, bp = f(C, C) * 0.1, f(C)
dadj = f(BT, HS * N)
dyn = f(B, HT, T * HS) * 0.1
tmpl = torch.arange(12, device=dev, dtype=torch.float32) / 12
Rr, out = ops.hypertem_fwd(X, G, Wbt, bbt)
X2, dO2, out2 = X.view(-1, C), dO.view(-1, C), out.view(-1, C)
dGp = torch.empty(B, N, 12, 12, device=dev)
dbn = torch.zeros(N, C, device=dev)
c, s = ops.cap_route_fwd(X, Wp, bp, dadj, HS, R)
v, Ht, Rt = ops.cap_cross_fwd(s, dyn, tmpl, B, T, HS, HT)
rec = ops.cap_rec_fwd(c, v, N, C)
dc1, dv = ops.cap_rec_bwd(dO2, c, v)
dS, ddyn = ops.cap_cross_bwd(dv, s, Rt, Ht, dyn, tmpl, B, T, HS, HT)

Wo, bo = f(1, C) * 0.2, f(1)
W3, b3 = f(HS, C) * 0.2, f(HS)
Wi, bi = f(C, 1), f(C)
src = f(B * T * N, 3)
mask = (torch.rand(B * T * N, device=dev) > 0.25).float()
prob = torch.softmax(f(B * T * N, HS), -1)
stats = torch.zeros(8, device=dev)
sws = ops.tail_sws(B * T * N, dev)
lc = torch.tensor([3, 1, 0, 4, 2, 9, 8, 7, 6, 5], dtype=torch.int32, device=dev)
nums = torch.tensor([8000, 8320], dtype=torch.int32, device=dev)
na, nr = torch.rand(B * T * N, device=dev), torch.rand(B * T * N, device=dev)

CASES = {
    "hypertem_fwd": lambda: ops.hypertem_fwd(X, G, Wbt, bbt),
    "hypertem_bwd": lambda: ops.hypertem_bwd(dO, out, X, G, Wbt, dG=dGp, want_dbias=False),
    "hypertem_bwd_wgrad": lambda: ops.hypertem_bwd_wgrad(dO, out, X, G, Wbt, Rr, dG=dGp),
    "wgrad_time_dpre_cs": lambda: ops.wgrad(Rr.view(-1, C), dO2, MODE_TIME, BT, N, D2=out2, pro=PRO_DPRE, colsum_d=True),
    "wgrad_node_dpre": lambda: ops.wgrad(rec, dO2, MODE_NODE, BT, N, D2=out2, pro=PRO_DPRE),
    "wgrad_shared_cs": lambda: ops.wgrad(dO2, X2, MODE_SHARED, BT, N, colsum_a=True),
    "apply_node_fwd": lambda: ops.apply(rec, Wn, MODE_NODE, BT, N, bias=bn, resid=X2, epi=EPI_RES_LRELU),
    "apply_node_dpre": lambda: ops.apply(dO2, Wn, MODE_NODE, BT, N, A2=out2, transw=True, pro=PRO_DPRE, colsum=True),
    "apply_wgrad_node": lambda: ops.apply_wgrad(dO2, out2, rec, Wn, MODE_NODE, BT, N),
    "apply_wgrad_time": lambda: ops.apply_wgrad(dO2, out2, X2, Wbt, MODE_TIME, BT, N),
    "linear_bwd_shared": lambda: ops.linear_bwd(dO2, X2, Wp, dO2, out2),
    "apply_shared_dx": lambda: ops.apply(dO2, Wp, MODE_SHARED, BT, N, resid=dO2, resid2=out2, epi=EPI_ADD_DPRE),
    "cap_route_fwd": lambda: ops.cap_route_fwd(X, Wp, bp, dadj, HS, R),
    "cap_cross_fwd": lambda: ops.cap_cross_fwd(s, dyn, tmpl, B, T, HS, HT),
    "cap_rec_fwd": lambda: ops.cap_rec_fwd(c, v, N, C),
    "cap_rec_bwd": lambda: ops.cap_rec_bwd(dO2, c, v),
    "cap_cross_bwd": lambda: ops.cap_cross_bwd(dv, s, Rt, Ht, dyn, tmpl, B, T, HS, HT),
    "cap_route_bwd": lambda: ops.cap_route_bwd(X, Wp, bp, c, dc1, dS),
    "chain_ht_ht": lambda: ops.hypertem_chain_fwd(X, [(G, Wbt, bbt), (G, Wbt, bbt)]),
    "chain_ht": lambda: ops.hypertem_chain_fwd(X, [(G, Wbt, bbt)]),
    "copy_A": lambda: X2.clone(),
    "tail_mae": lambda: ops.tail_mae(X2, Wo, bo, src, 3, mask, 146.0, 230.0, 0.0, sws),
    "tail_kl": lambda: ops.tail_kl(X2, W3, prob, c, N, 0.1, sws),
    "stats_fold": lambda: ops.stats_fold(sws, stats),
    "rowdot_softmax": lambda: ops.rowdot(X2, W3, b3, softmax=True, want_label=True),
    "rowdot_softmax_nolabel": lambda: ops.rowdot(X2, W3, b3, softmax=True),
    "rowdot_j1": lambda: ops.rowdot(X2, Wo, bo),
    "lin_in": lambda: ops.lin_in(src, 3, 1, Wi, bi, C),
    "mask_adaptive": lambda: ops.mask_adaptive(*ops.mask_labels(prob), lc, nums, na, nr, True, 1),
}


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1000)
    return best


flt = sys.argv[1] if len(sys.argv) > 1 else ""
if flt == "capgen":           # cap_route_fwd: third generation (default), its <= 80 VGPR variant, the second generation
    from gptst_amd import _C
    for lag in (1, 2, 3, 4, 6, 8, 10, 12):
        _C.lib().call("gptst_tune", 22, lag)
        print("cap_route_fwd fwd4 lag %2d %7.2f us" % (lag, bench(CASES["cap_route_fwd"])))
    _C.lib().call("gptst_tune", 22, 0)
    for nm, kv in (("fwd4", ()), ("fwd2", ((20, 1),))):
        for k, v in kv:
            _C.lib().call("gptst_tune", k, v)
        print("cap_route_fwd %-10s %7.2f us" % (nm, bench(CASES["cap_route_fwd"])))
        for k, v in kv:
            _C.lib().call("gptst_tune", k, 0)
    sys.exit(0)
if flt == "tailsweep":
    from gptst_amd import _C
    for nb in (256, 512, 1024, 2048, 4080):
        _C.lib().call("gptst_tune", 6, nb)
        print("TL_NB", nb, "tail_mae %.2f us  tail_kl %.2f us" % (bench(CASES["tail_mae"]), bench(CASES["tail_kl"])))
    sys.exit(0)
tot = 0.0
for k, fn in CASES.items():
    if flt in k:
        t = bench(fn)
        print("%-20s %7.2f us" % (k, t))
