#!/usr/bin/env python
"""Kernel-trace summary (tools/prof.sh -> gpurun_out/<tag>_trace.txt, the rocprofv3 --kernel-trace --stats run of bench.py under hipGraph replay) ->
profiles/graph_kernel_us.json: average launch duration per kernel symbol, stamped with the hash of the kernel sources it was collected on.  bench.py puts
these durations (`avg_us_graph`) beside its own eager event-pair timings in the `roofline` object when the hash matches the running sources.
    python tools/trace_to_json.py gpurun_out/r06_trace.txt profiles/graph_kernel_us.json"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402

src, out = sys.argv[1:3]
res = {}
for line in open(src):
    m = re.match(r"^(.*\S)\s+\[(\d+),(\d+)\]\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if m:
        res["%s [%s,%s]" % (m.group(1).strip(), m.group(2), m.group(3))] = {"calls": int(m.group(4)), "avg_us": float(m.group(5)), "us_per_step": float(m.group(7))}
json.dump({"note": "rocprofv3 --kernel-trace --stats of `bench.py --steps 60 --warmup 4` under hipGraph replay (tools/prof.sh); avg_us per launch",
           "source": os.path.basename(src), "kernel_src_sha": kernel_source_hash(), "kernels": res}, open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
