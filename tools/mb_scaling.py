#!/usr/bin/env python
"""How the hot kernels' time scales with the batch (B = 8 .. 128 at N = 170, C = 64): flat = latency chain, linear = throughput.
usage (GPU box): python tools/mb_scaling.py"""
import sys
sys.path.insert(0, '.')
import torch
from gptst_amd import ops
from gptst_amd.ops import MODE_TIME, MODE_NODE, PRO_DPRE, EPI_RES_LRELU

dev = 'cuda:0'
T, N, C, HS, R = 12, 170, 64, 10, 2
f = lambda *s: torch.randn(*s, device=dev)


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


print("%4s %10s %10s %10s %10s %10s %10s" % ("B", "ht_fwd", "ht_bwd", "wgrad_t", "applywg_n", "route_fwd", "route_bwd"))
for B in (8, 16, 32, 64, 128):
    BT = B * T
    X, dO = f(B, T, N, C), f(B, T, N, C)
    G = f(N, 12, 12) * 0.1
    Wbt, bbt = f(BT, C, C) * 0.1, f(BT, C)
    Wn = f(N, C, C) * 0.1
    Wp, bp = f(C, C) * 0.1, f(C)
    dadj = f(BT, HS * N)
    Rr, out = ops.hypertem_fwd(X, G, Wbt, bbt)
    dGp = torch.empty(B, N, 12, 12, device=dev)
    c, s = ops.cap_route_fwd(X, Wp, bp, dadj, HS, R)
    dc1, dS = torch.randn_like(c), torch.randn_like(s)
    X2, dO2, out2 = X.view(-1, C), dO.view(-1, C), out.view(-1, C)
    t = [bench(lambda: ops.hypertem_fwd(X, G, Wbt, bbt)),
         bench(lambda: ops.hypertem_bwd(dO, out, X, G, Wbt, dG=dGp, want_dbias=False)),
         bench(lambda: ops.wgrad(Rr.view(-1, C), dO2, MODE_TIME, BT, N, D2=out2, pro=PRO_DPRE, colsum_d=True)),
         bench(lambda: ops.apply_wgrad(dO2, out2, X2, Wn, MODE_NODE, BT, N)),
         bench(lambda: ops.cap_route_fwd(X, Wp, bp, dadj, HS, R)),
         bench(lambda: ops.cap_route_bwd(X, Wp, bp, c, dc1, dS))]
    print("%4d " % B + " ".join("%10.1f" % v for v in t))
