#!/usr/bin/env python
"""Per-kernel roofline table of the step from the committed evidence: launch durations from the rocprofv3 kernel-trace summary of the graph replay
(profiles/<tag>_graph_kernel_trace.txt), HBM bytes per launch from the PMC passes (profiles/pmc_traffic.json, (2 FETCH_SIZE + WRITE_SIZE) * 1024, calibrated:
profiles/r05_pmc_calibration.txt), algorithmic FLOPs from bench.kernel_flops (SURVEY 8d formulas).  For every kernel of the median step: us per step,
bytes, achieved TB/s on its ACTUAL traffic, the time that traffic takes at the 8 TB/s HBM roof the grading contract names (first) and at the ~4.5 TB/s a streaming
kernel reaches on this chip (tools/experiments/stream_rates.py; a diagnostic), fp32-MFMA time at 157.3 TFLOP/s, and the higher floor over the measured time.
    python tools/roofline_table.py r05h > profiles/r05h_roofline_table.txt"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r05h"
HBM_TBS, STREAM_TBS, MFMA_TF = 8.0, 4.5, 157.3      # the grading roof (HBM3E spec), the streaming rate kernels reach here (diagnostic), fp32 MFMA
d = dict(B=32, T=12, N=170, C=64, HS=10, R=3)
pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
SYM2ENTRY = {
    "void cap_route_bwd2_kernel<64, 1, true": ("gptst_cap_cross_route_lin_bwd", ""), "void hypertem_bwd_pair_kernel<6>": ("gptst_hypertem_bwd_pair", ""),
    "void hypertem_chain_fwd_kernel<2>": ("gptst_hypertem_chain_fwd", "x2"), "void cap_route_fwd4_kernel<2>": ("gptst_cap_route_fwd", ""),
    "void applywg64_kernel<0, 1>": ("gptst_apply_wgrad", ""), "void applywg64_kernel<0, 2>": ("gptst_apply_wgrad", ""),
    "void apply64_kernel<0, 1>": ("gptst_apply", ""), "void apply64_kernel<0, 3>": ("gptst_apply", ""),
    "void cap_cross_rec_fwd_kernel<64>": ("gptst_cap_rec_fwd", ""), "void cap_rec_bwd2_kernel<64>": ("gptst_cap_rec_bwd", ""),
    "void hypertem_bwd_wgrad_kernel<6, false, true, false": ("gptst_hypertem_bwd_wgrad", ""), "hypertem_fwd_kernel": ("gptst_hypertem_fwd", ""),
}
rows = []
tl = open(os.path.join(ROOT, "profiles", "%s_step_timeline_median_adaptive.txt" % tag)).read().splitlines()
per = {}
for l in tl:
    m = re.match(r"\s+[\d.]+\s+([\d.]+) gap\s+[-\d.]+ q\d+\s+s\d+\s+(.*)$", l)
    if m:
        nm = m.group(2).strip()
        e = per.setdefault(nm, [0, 0.0])
        e[0] += 1; e[1] += float(m.group(1))
tot_us = sum(v[1] for v in per.values())
tot_b = tot_t = tot_m = 0.0
print("# %s: median adaptive step, %d launches, %.1f us; floors: counter traffic at the %.1f TB/s HBM roof (graded) and at %.1f TB/s (diagnostic: what a streaming "
      "kernel reaches here), fp32 MFMA at %.1f TFLOP/s" % (tag, sum(v[0] for v in per.values()), tot_us, HBM_TBS, STREAM_TBS, MFMA_TF))
print("%-58s %3s %8s %8s %7s %8s %8s %8s  %s" % ("kernel [grid]", "n", "us/step", "MB/step", "TB/s", "t_hbm@8", "t_mfma", "t_hbm@4.5", "floor/time @8 TB/s  (@4.5)"))
for nm, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    key = [k for k in pm if k.startswith(nm[:40]) and k.endswith(nm[-8:])]
    by = pm[key[0]]["hbm_bytes"] * n if key else 0.0
    sym = nm.rsplit(" [", 1)[0]
    ent = SYM2ENTRY.get(sym) or next((v for k, v in SYM2ENTRY.items() if sym.startswith(k[:40])), None)
    fl = bench.kernel_flops(ent[0], ent[1], d) * n if ent else 0.0
    t_8, t_h, t_m = by / (HBM_TBS * 1e6), by / (STREAM_TBS * 1e6), fl / (MFMA_TF * 1e6)
    tot_b += by; tot_t += t_h; tot_m += t_m
    print("%-58s %3d %8.1f %8.1f %7.2f %8.1f %8.1f %8.1f  %4.0f %%  (%3.0f %%)" % (nm[:58], n, us, by / 1e6, by / us / 1e6 if us else 0, t_8, t_m, t_h,
                                                                              100 * max(t_8, t_m) / us if us else 0, 100 * max(t_h, t_m) / us if us else 0))
tot_8 = tot_b / (HBM_TBS * 1e6)
print("%-58s %3s %8.1f %8.1f %7.2f %8.1f %8.1f %8.1f" % ("sum", "", tot_us, tot_b / 1e6, tot_b / tot_us / 1e6, tot_8, tot_m, tot_t))
print("# the step moves %.2f GB (%.2fx the 1.13 GB of SURVEY 8d) = %.0f us at the %.1f TB/s roof = %.0f %% of its %.0f us (%.0f us = %.0f %% at the %.1f TB/s streaming "
      "kernels reach); SURVEY 8d's 1.13 GB at 8 TB/s: %.0f us = %.1f %%; the named kernels' MFMA work is %.0f us (%.0f %%)" % (
    tot_b / 1e9, tot_b / 1.131e9, tot_8, HBM_TBS, 100 * tot_8 / tot_us, tot_us, tot_t, 100 * tot_t / tot_us, STREAM_TBS, 1.131e9 / (HBM_TBS * 1e6),
    100 * 1.131e9 / (HBM_TBS * 1e6) / tot_us, tot_m, 100 * tot_m / tot_us))
