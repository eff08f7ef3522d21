cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CMD="python bench.py --steps 60 --warmup 4 --no-cpu-baseline --no-kernel-timing"
PROF_LINES=5 tools/prof.sh r02w_trace 64 $CMD > /dev/null
db=$(ls /tmp/prof_r02w_trace/*.db | head -1)
python tools/timeline.py $db > gpurun_out/r02w_timeline.txt
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do PMC_LINES=60 tools/pmc.sh r02w $c $CMD > /dev/null; done
python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r02w_bench.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02w_bench_driver_cmd.json
head -4 gpurun_out/r02w_trace.txt; tail -3 gpurun_out/r02w_timeline.txt
python -c "
import json
for f in ('r02w_bench','r02w_bench_driver_cmd'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'])
"
