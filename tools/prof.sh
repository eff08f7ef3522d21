#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <name> <steps-divisor> <command...>
# runs the command under rocprofv3 --kernel-trace --stats and prints/saves the per-kernel summary (gpurun_out/<name>.txt)
name=$1; div=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- bash -c "cd $root && $*" ) > $root/gpurun_out/$name.log 2>&1
db=$(ls /tmp/prof_$name/*.db | head -1)
python $root/tools/rocpd_summary.py $db $div > $root/gpurun_out/$name.txt
head -${PROF_LINES:-45} $root/gpurun_out/$name.txt
