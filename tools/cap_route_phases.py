#!/usr/bin/env python
"""Per-phase s_memtime stamps of cap_route_fwd2_kernel (workgroup 5, thread 0) at the bench shape — needs the -DGPTST_DEBUG build:
    GPTST_EXTRA_HIPCC_FLAGS=-DGPTST_DEBUG python -m gptst_amd.build --force;  python tools/cap_route_phases.py > profiles/<tag>_cap_route_fwd2_phases.txt
Cycles are shader-clock cycles of ONE workgroup's critical path (every phase ends in a workgroup barrier); the launch itself is timed
under graph replay next to it."""
import ctypes
import sys
sys.path.insert(0, '.')
import torch
from gptst_amd import ops, _C
dev = 'cuda:0'
T, N, C, HS, R = 12, 170, 64, 10, 2
dll = ctypes.CDLL(_C.LIB_PATH)
names = ["stage W + zero V (X tile requested)", "P = squash(X Wp^T + bp)  [MFMA 16x16x4, 11 row tiles]", "zero b / c", "c0 = softmax_h(dadj)",
         "S = c0 . P  [type 1, MFMA]", "v0 = squash(S)", "r0: c = softmax_h(b)", "r0: S = c . P", "r0: v = squash(v0 (.) S)",
         "r1: b += v . P^T  [type 2, MFMA]", "r1: c = softmax_h(b)", "r1: S = c . P", "r1: v = squash(v0 (.) S)", "b += v . P^T",
         "c = softmax_h(b + dadj) -> c_out", "S = c . P", "s -> s_out (post, after the last stamp)"]
for B in (8, 32):
    torch.manual_seed(0)
    X = torch.randn(B, T, N, C, device=dev); Wp = torch.randn(C, C, device=dev) * 0.1; bp = torch.randn(C, device=dev)
    dadj = torch.randn(B * T, HS * N, device=dev)
    dll.gptst_tune2(0)
    for _ in range(3):
        ops.cap_route_fwd(X, Wp, bp, dadj, HS, R)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.cap_route_fwd(X, Wp, bp, dadj, HS, R)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50)
    dll.gptst_tune2(99)
    for _ in range(50):
        ops.cap_route_fwd(X, Wp, bp, dadj, HS, R)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    dll.gptst_cap_ts(buf)
    ts = list(buf)
    n = max(i for i in range(40) if ts[i] > 0)
    tot = ts[n] - ts[0]
    print("B = %d (%d workgroups of 512 threads, %s): %.1f us per launch under graph replay; workgroup 5: %d cycles from its first to its last stamp"
          % (B, B * T, "<= 1 per CU" if B * T <= 256 else "2 per CU on half of the CUs", best, tot))
    for i in range(1, n + 1):
        print("   %-58s %7d cycles  %5.1f %%" % (names[i - 1] if i - 1 < len(names) else "?", ts[i] - ts[i - 1], 100.0 * (ts[i] - ts[i - 1]) / tot))
    dll.gptst_tune2(0)
