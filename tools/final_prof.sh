#!/bin/bash
# usage (GPU box, repo root): tools/final_prof.sh <tag>   e.g. r03z
# The round's evidence set, all from ONE command line: rocprofv3 kernel trace + per-step timeline, four PMC passes (separate runs, kernel
# trace only), profiles-ready pmc_traffic.json (stamped with the kernel-source hash bench.py checks), and the bench JSON lines.
tag=${1:-r03}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CMD="python bench.py --steps 60 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-module-path"
PROF_LINES=5 tools/prof.sh ${tag}_trace auto $CMD > /dev/null
db=$(ls /tmp/prof_${tag}_trace/*.db | head -1)
python tools/timeline.py $db > gpurun_out/${tag}_timeline.txt
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do PMC_LINES=60 tools/pmc.sh ${tag} $c $CMD --no-graph > /dev/null; done
python tools/pmc_to_json.py gpurun_out/pmc_${tag}_FETCH_SIZE.txt gpurun_out/pmc_${tag}_WRITE_SIZE.txt gpurun_out/${tag}_pmc_traffic.json
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json      # so that the bench runs below report `traffic`
python tools/trace_to_json.py gpurun_out/${tag}_trace.txt gpurun_out/${tag}_graph_kernel_us.json
cp gpurun_out/${tag}_graph_kernel_us.json profiles/graph_kernel_us.json   # ... and `avg_us_graph` (the graph-replay durations beside the eager event pairs)
python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/${tag}_bench.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_driver_cmd.json
head -4 gpurun_out/${tag}_trace.txt; tail -3 gpurun_out/${tag}_timeline.txt
python -c "
import json
for f in ('${tag}_bench','${tag}_bench_driver_cmd'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'))
"
