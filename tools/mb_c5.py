#!/usr/bin/env python
"""Micro-benchmarks at the BASELINE configs[4] shape (B=32,T=12,N=4096,C=128): hyperTem forward/backward as whole-batch launches vs
sample chunks that keep the intermediate tensors inside the 256 MB Infinity Cache.  usage (GPU box): python tools/mb_c5.py [chunk ...]"""
import sys
sys.path.insert(0, '.')
import torch
from gptst_amd import ops
from gptst_amd.ops import MODE_TIME, PRO_DPRE, EPI_RES_LRELU

dev = 'cuda:0'
B, T, N, C = 32, 12, 4096, 128
torch.manual_seed(0)
f = lambda *s: torch.randn(*s, device=dev)
X, dO = f(B, T, N, C), f(B, T, N, C)
G = f(N, 12, 12) * 0.1
Wbt, bbt = f(B * T, C, C) * 0.1, f(B * T, C)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000


def fwd(cb):
    outs = []
    for b0 in range(0, B, cb):
        x = X[b0:b0 + cb]
        R = ops.tmix(x, G)
        out = ops.apply(R.view(-1, C), Wbt[b0 * T:(b0 + cb) * T], MODE_TIME, cb * T, N, bias=bbt[b0 * T:(b0 + cb) * T], resid=x.view(-1, C),
                        epi=EPI_RES_LRELU)
        outs.append((R, out))
    return outs


def bwd(cb, saved):
    for i, b0 in enumerate(range(0, B, cb)):
        R, out = saved[i]
        x, do = X[b0:b0 + cb], dO[b0:b0 + cb].view(-1, C)
        w = Wbt[b0 * T:(b0 + cb) * T]
        dR = ops.apply(do, w, MODE_TIME, cb * T, N, A2=out, transw=True, pro=PRO_DPRE)
        dWb, ns = ops.wgrad(R.view(-1, C), do, MODE_TIME, cb * T, N, D2=out, pro=PRO_DPRE, colsum_d=True)
        dx, dG = ops.tmix_bwd(dR.view(cb, T, N, C), x, G, do.view(cb, T, N, C), out.view(cb, T, N, C))


chunks = [int(a) for a in sys.argv[1:]] or [32, 8, 4, 2]
for cb in chunks:
    saved = fwd(cb)
    print("chunk %2d samples: fwd %8.1f us   bwd %8.1f us" % (cb, timeit(lambda: fwd(cb)), timeit(lambda: bwd(cb, saved))))
    del saved
