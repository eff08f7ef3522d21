#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh <name> <counter> <command...>
# One PMC counter per pass, kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Prints / saves per-kernel average counter values: gpurun_out/pmc_<name>_<counter>.txt  (value unit: KB for *_SIZE)
name=$1; ctr=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${name}_$ctr -o $name -- bash -c "cd $root && $*" ) > $root/gpurun_out/pmc_${name}_$ctr.log 2>&1
db=$(ls /tmp/pmc_${name}_$ctr/*.db | head -1)
python - > $root/gpurun_out/pmc_${name}_$ctr.txt <<PY
import sqlite3, re
db = sqlite3.connect("$db")
rows = db.execute("select kernel_name, grid_size_x/workgroup_size_x, grid_size_y, value, duration from counters_collection where counter_name='$ctr'").fetchall()
agg = {}
for n, bx, gy, v, d in rows:
    k = (re.sub(r"\(.*$", "", n)[:60], bx, gy)
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += v; a[2] += d
print("# $ctr per kernel (avg per launch), same command as the kernel trace; durations under PMC collection are not representative")
print("%-72s %8s %14s" % ("kernel [blocks_x, grid_y]", "calls", "avg_$ctr"))
for k, (c, v, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-72s %8d %14.1f" % ("%s [%d,%d]" % k, c, v / c))
PY
head -${PMC_LINES:-14} $root/gpurun_out/pmc_${name}_$ctr.txt
