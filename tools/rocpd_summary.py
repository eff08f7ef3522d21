#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace): per kernel (name, grid) count / avg / total.
usage: python tools/rocpd_summary.py results.db [steps|auto] > profiles/xxx.txt
The per-step column divides by the number of optimiser steps IN THE TRACE = launches of adam_kernel (warm-up and capture steps included);
a numeric `steps` argument is only used when the trace holds no adam_kernel (micro-benchmarks)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "auto" else 1.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_c = "name" if "name" in cols else "kernel_name"
rows = db.execute("select %s, grid_x, grid_y, workgroup_x, (end-start) from kernels" % name_c).fetchall()
agg = {}
for n, gx, gy, wx, dur in rows:
    n = re.sub(r"\(.*$", "", n)
    k = (n, gx // max(wx, 1), gy)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += dur
tot = sum(a[1] for a in agg.values())
nadam = sum(c for (n, _, _), (c, _) in agg.items() if n.startswith("adam_kernel"))
if nadam:
    steps = float(nadam)
print("total kernel time %.1f us over %d dispatches; per step (/%g): %.1f us" % (tot / 1e3, len(rows), steps, tot / 1e3 / steps))
print("%-70s %10s %8s %10s %10s %6s" % ("kernel [blocks_x, grid_y]", "calls", "avg_us", "total_us", "us/step", "%"))
for (n, bx, gy), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-70s %10d %8.1f %10.1f %10.1f %6.1f" % (("%s [%d,%d]" % (n[:52], bx, gy)), c, t / c / 1e3, t / 1e3, t / 1e3 / steps, 100 * t / tot))
