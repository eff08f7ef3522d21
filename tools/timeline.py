#!/usr/bin/env python
"""Timeline of ONE step from a rocprofv3 rocpd sqlite database (kernel-trace): per-kernel start offset, duration, and the gap
to the previous kernel on the same queue (steps are delimited by `marker` kernel name, default adam_kernel).  Which step: by default the
MEDIAN one (by span) among the steps with the most launches — the adaptive-mask + KL phase of a bench run, not a random-phase warm-up step
and not a tie-path outlier; `which_from_end` = an integer picks that step counted from the end instead.
usage: python tools/timeline.py results.db [marker] [median|which_from_end] > gpurun_out/timeline.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
which = sys.argv[3] if len(sys.argv) > 3 else "median"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("# columns:", cols)
name_c = "name" if "name" in cols else "kernel_name"
qc = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
sc = "stream_id" if "stream_id" in cols else None
sel = "select %s, grid_x, grid_y, workgroup_x, start, end%s%s from kernels order by start" % (
    name_c, (", " + qc) if qc else "", (", " + sc) if sc else "")
rows = db.execute(sel).fetchall()
ends = [i for i, r in enumerate(rows) if marker in r[0]]
if len(ends) < 5:
    print("not enough steps", len(ends)); sys.exit(0)
if which == "median":
    cand = [(a + 1, b + 1) for a, b in zip(ends[:-1], ends[1:])]
    cnt = lambda c: sum(1 for r in rows[c[0]:c[1]] if not r[0].startswith("__amd_rocclr"))      # (the replay's H2D copy rides in every 4th step)
    hist = {}
    for c in cand:
        hist[cnt(c)] = hist.get(cnt(c), 0) + 1
    nmax = max(n for n, k in hist.items() if k >= max(4, len(cand) // 10))        # the largest launch count that is a regular step, not a one-off
    cand = sorted((c for c in cand if cnt(c) == nmax), key=lambda c: rows[c[1] - 1][5] - rows[c[0]][4])
    i0, i1 = cand[len(cand) // 2]
    note = "median by span of the %d steps with %d launches" % (len(cand), nmax)
else:
    which = int(which)
    i0, i1 = ends[-which - 1] + 1, ends[-which] + 1
    note = "step %d from the end" % which
step = rows[i0:i1]
t0 = step[0][4]
print("# step (%s): %d kernels, span %.1f us" % (note, len(step), (step[-1][5] - t0) / 1e3))
last_end = {}
busy = {}
for r in step:
    n = re.sub(r"\(.*$", "", r[0])[:44]
    q = r[6] if qc else 0
    s = r[7] if (qc and sc) else (r[6] if sc else 0)
    gap = (r[4] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r[5]
    busy[q] = busy.get(q, 0.0) + (r[5] - r[4]) / 1e3
    print("%9.1f %7.1f gap %6.1f q%-3s s%-3s %s [%d,%d]" % ((r[4] - t0) / 1e3, (r[5] - r[4]) / 1e3, gap, q, s, n, r[1] // max(r[3], 1), r[2]))
print("# busy us per queue:", {k: round(v, 1) for k, v in busy.items()})
# inter-step statistics over the whole run: period = marker end -> next marker end; idle = marker end -> next kernel start
per, idle, span = [], [], []
for a, b in zip(ends[:-1], ends[1:]):
    per.append((rows[b][5] - rows[a][5]) / 1e3)
    idle.append((rows[a + 1][4] - rows[a][5]) / 1e3)
    span.append((rows[b][5] - rows[a + 1][4]) / 1e3)
if per:
    med = lambda v: sorted(v)[len(v) // 2]
    print("# steps %d: median period %.1f us, median span %.1f us, median idle between steps %.1f us" % (len(per), med(per), med(span), med(idle)))
