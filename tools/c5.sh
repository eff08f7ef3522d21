#!/bin/bash
# usage (GPU box, repo root): tools/c5.sh <tag> [batch] [extra bench args] — BASELINE configs[4] shape on one GPU, prints steps/s and the kernel breakdown
tag=$1; b=${2:-32}; shift 2
mkdir -p gpurun_out/c5
python bench.py --nodes 4096 --hidden 128 --batch $b --steps 6 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/c5/$tag.json 2> gpurun_out/c5/$tag.err || tail -5 gpurun_out/c5/$tag.err
python - <<PY
import json
d = json.load(open("gpurun_out/c5/$tag.json"))
print("steps/s %.3f  ms/step %.2f  kernel sum (eager) %.1f us" % (d["value"], d["ms_per_step"], d.get("kernel_time_sum_us_per_step_eager", 0)))
for k, v in list(d.get("kernel_breakdown_us_per_step", {}).items())[:${C5_LINES:-16}]:
    print("%-52s %9.1f" % (k, v))
PY
