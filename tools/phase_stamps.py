#!/usr/bin/env python
"""Per-phase / per-workgroup wall-clock stamps of a kernel (needs the -DGPTST_STAMPS build of the library:
    GPTST_EXTRA_HIPCC_FLAGS=-DGPTST_STAMPS python -m gptst_amd.build --force && cp gpt-st_amd/lib/libgptst_hip.so gpt-st_amd/lib/libgptst_hip_stamps.so
    python -m gptst_amd.build --force      # the product build again
    GPTST_LIB=$PWD/gpt-st_amd/lib/libgptst_hip_stamps.so python tools/phase_stamps.py cap_route_bwd      (GPU box)
Prints, for the stamped workgroups, the time between consecutive stamps, and the launch's schedule: start / end of every workgroup."""
import ctypes
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from gptst_amd import ops, _C

dev = 'cuda:0'
B, T, N, C, HS, HT, R = 32, 12, 170, 64, 10, 16, 2
BT = B * T
torch.manual_seed(0)
f = lambda *s: torch.randn(*s, device=dev)
X, dO = f(B, T, N, C), f(B * T * N, C)
Wp, bp = f(C, C) * 0.1, f(C)
dadj = f(BT, HS * N)
dyn = f(B, HT, T * HS) * 0.1
tmpl = torch.arange(12, device=dev, dtype=torch.float32) / 12
c, s = ops.cap_route_fwd(X, Wp, bp, dadj, HS, R)
v, Ht, Rt = ops.cap_cross_fwd(s, dyn, tmpl, B, T, HS, HT)
dc1, dv = ops.cap_rec_bwd(dO, c, v)
src3 = f(B, T, N, 3)
mask1 = (torch.rand(B * T * N, device=dev) > 0.25).float()
w_in, b_in = f(C, 1), f(C)
Gt = f(N, 12, 12) * 0.1
Wbt, bbt = f(BT, C, C) * 0.1, f(BT, C)
o_ei, ab_ei, wv_ei = ops.encin_ht1_fwd(src3, 1, mask1, -1.5, w_in, b_in, Gt, Wbt, bbt)
def _pair_case():
    g = lambda *sh: torch.randn(*sh, device=dev)
    x0 = g(B, T, N, C)
    G0, G1 = g(N, T, T) * 0.2, g(N, T, T) * 0.2
    W0, W1, b0, b1 = g(BT, C, C) * 0.1, g(BT, C, C) * 0.1, g(BT, C), g(BT, C)
    R0, x1 = ops.hypertem_fwd(x0, G0, W0, b0)
    R1, x2 = ops.hypertem_fwd(x1, G1, W1, b1)
    dpre1 = g(B, T, N, C)
    dG1, dG0 = torch.empty(B, N, T, T, device=dev), torch.empty(B, N, T, T, device=dev)
    return lambda: ops.hypertem_bwd_pair(dpre1, x1, G1, W1, R1, x0, G0, W0, R0, dG1, dG0, torch.zeros(B, device=dev))


CASES = {
    "ht_bwd_pair": ("hypertem", _pair_case(),
                    ["stage 1: loads -> first time-step group staged in LDS", "stage 1: dR = dPre W^T (3 groups of 4 time steps, MFMA)", "stage 1: dX (VALU) + write-through stores",
                     "stage 1: dG (MFMA)", "drain the write-through stores, count up", "stage 2: sign / graph / W loads -> first group staged", "stage 2: dR", "stage 2: dX + stores", "stage 2: dG"]),
    "encin_fwd": ("encin", lambda: ops.encin_ht1_fwd(src3, 1, mask1, -1.5, w_in, b_in, Gt, Wbt, bbt),
                  ["w W_bt, bi W_bt partials", "alpha / beta / m per node (G rows, flow, mask) + barrier", "rows out"]),
    "encin_bwd": ("encin", lambda: ops.encin_ht1_bwd(dO, src3, mask1, -1.5, w_in, b_in, Wbt, ab_ei, wv_ei),
                  ["pass over dPre: A, Bv, Cv, Mv, row dots", "fold slots", "dWb row + projection partials", "dG rows"]),
    "cap_route_bwd_roles": ("capmfma", lambda: ops.cap_cross_route_bwd(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, B, T, HS, HT, flags=torch.zeros(4 * B, device=dev)),
                      ["(cross-time role: whole launch) / routing role: -", "stage Wp (+ X tile requested)", "Y = X Wp^T + bp tiles -> LDS", "zero c / dc, stage c, dc1, wait for dS",
                       "node tiles: U, dlogit, dP, squash backward, dY rows out"]),
    "cap_route_lin_bwd_roles": ("capmfma", lambda: ops.cap_cross_route_lin_bwd(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, dO, None, True, B, T, HS, HT, flags=torch.zeros(4 * B, device=dev)),
                      ["(cross-time role: whole launch) / routing role: -", "stage Wp (+ X tile requested)", "Y = X Wp^T + bp tiles -> LDS", "zero c / dc, stage c, dc1, wait for dS",
                       "node tiles: U, dlogit, dP, squash backward (dY stays in LDS)", "barrier, Wp -> LDS, barrier", "dX tiles = dY Wp + residual branch, rows out", "dWp / dbp partial of the (b,t)"]),
    "cap_route_bwd": ("capmfma", lambda: ops.cap_cross_route_bwd(X, Wp, bp, c, dc1, dv, s, Rt, Ht, dyn, tmpl, B, T, HS, HT),
                      ["cross-time backward prologue (replicated per (b,t))", "stage Wp (+ X tile requested)", "Y = X Wp^T + bp tiles -> LDS", "zero c / dc, stage c, dc1",
                       "node tiles: U, dlogit, dP, squash backward, dY rows out"]),
}
name = sys.argv[1] if len(sys.argv) > 1 else "cap_route_bwd"
unit, fn, names = CASES[name]
dll = ctypes.CDLL(_C.LIB_PATH)
for _ in range(5):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    fn()
e1.record(); torch.cuda.synchronize()
ph = (ctypes.c_longlong * (8 * 32))()
wg = (ctypes.c_longlong * (2048 * 2))()
assert getattr(dll, "gptst_stamps_" + unit)(ph, wg) == 0
nwg = BT + (4 * B if name.endswith('roles') else 0)
if len(sys.argv) > 2:            # launch geometry given (node halves, r06): workgroup count, printed in chunks of 64
    nwg = int(sys.argv[2])
if name == "ht_bwd_pair":
    nwg = 352 + 2 * BT
ph = np.array(ph).reshape(8, 32); wg = np.array(wg).reshape(2048, 2)[:nwg]
t0 = wg[:, 0].min()
print("%s: %.2f us per launch (50 back to back); workgroups start %.2f .. %.2f us, end %.2f .. %.2f us after the first start; mean duration %.2f us"
      % (name, e0.elapsed_time(e1) * 20, (wg[:, 0].min() - t0) * 0.01, (wg[:, 0].max() - t0) * 0.01, (wg[:, 1].min() - t0) * 0.01, (wg[:, 1].max() - t0) * 0.01,
         (wg[:, 1] - wg[:, 0]).mean() * 0.01))
for lo, hi in (((0, 352), (352, 352 + BT), (352 + BT, nwg)) if name == "ht_bwd_pair" else ((0, 128), (128, 256), (256, BT), (BT, BT + 4 * B), (BT + 4 * B, nwg))):
    if hi > lo and lo < nwg and hi <= nwg:
        w = wg[lo:hi]
        print("   workgroups %3d..%3d: end %.2f us (mean), duration %.2f us" % (lo, hi - 1, (w[:, 1].mean() - t0) * 0.01, (w[:, 1] - w[:, 0]).mean() * 0.01))
if len(sys.argv) > 2:
    for lo in range(0, nwg, 64):
        w = wg[lo:min(lo + 64, nwg)]
        print("   workgroups %3d..%3d: start %.2f end %.2f us (mean), duration %.2f us" % (lo, lo + len(w) - 1, (w[:, 0].mean() - t0) * 0.01, (w[:, 1].mean() - t0) * 0.01, (w[:, 1] - w[:, 0]).mean() * 0.01))
for sl, b in enumerate((5, 100, 200, 300, 261, 383)):
    if b >= BT or ph[sl, 0] == 0:
        continue
    n = max(i for i in range(32) if ph[sl, i] > 0)
    print("  workgroup %d: start %.2f, total %.2f us" % (b, (ph[sl, 0] - t0) * 0.01, (ph[sl, n] - ph[sl, 0]) * 0.01))
    for i in range(1, n + 1):
        print("     %-62s %6.2f us" % (names[i - 1] if i - 1 < len(names) else "?", (ph[sl, i] - ph[sl, i - 1]) * 0.01))
