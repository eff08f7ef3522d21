#!/bin/bash
# usage (GPU box, repo root): tools/pmc_multi.sh <name> "<CTR1 CTR2 ...>" <command...>
# ONE pass with several counters of the same block budget (SQ: 8 slots; MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only.
# Writes gpurun_out/pmcm_<name>.txt: per kernel [grid] the average of every counter per launch.
name=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmcm_${name} -o $name -- bash -c "cd $root && $*" ) > $root/gpurun_out/pmcm_${name}.log 2>&1
db=$(ls /tmp/pmcm_${name}/*.db | head -1)
python - > $root/gpurun_out/pmcm_${name}.txt <<PY
import sqlite3, re
db = sqlite3.connect("$db")
rows = db.execute("select kernel_name, grid_size_x/workgroup_size_x, grid_size_y, counter_name, value from counters_collection").fetchall()
agg, names = {}, []
for n, bx, gy, c, v in rows:
    k = (re.sub(r"\(.*$", "", n)[:56], bx, gy)
    if c not in names: names.append(c)
    a = agg.setdefault(k, {}).setdefault(c, [0, 0.0]); a[0] += 1; a[1] += v
print("%-66s %6s " % ("kernel [blocks_x, grid_y]", "calls") + " ".join("%20s" % c[-20:] for c in names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1].values())):
    c0 = max(x[0] for x in d.values())
    print("%-66s %6d " % ("%s [%d,%d]" % k, c0) + " ".join("%20.0f" % (d[c][1] / d[c][0]) if c in d else "%20s" % "-" for c in names))
PY
head -${PMC_LINES:-30} $root/gpurun_out/pmcm_${name}.txt
