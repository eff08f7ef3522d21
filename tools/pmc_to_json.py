#!/usr/bin/env python
"""Merge the per-kernel FETCH_SIZE / WRITE_SIZE summaries written by tools/pmc.sh into profiles/pmc_traffic.json.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports exactly half of a wide coalesced
read (MI355X_MICROARCH.md §HBM; re-calibrated here on a 268 MB copy: FETCH 130 872 KB, WRITE 261 719 KB for 261 719 KB moved)."""
import json
import re
import sys

fetch_txt, write_txt, out = sys.argv[1:4]


def parse(path):
    d = {}
    for line in open(path):
        m = re.match(r"^(.*\S)\s+\[(\d+),(\d+)\]\s+(\d+)\s+([\d.]+)\s*$", line)
        if m:
            d["%s [%s,%s]" % (m.group(1).strip(), m.group(2), m.group(3))] = float(m.group(5))
    return d


f, w = parse(fetch_txt), parse(write_txt)
res = {}
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, 0.0), w.get(k, 0.0)
    res[k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes": (2 * fk + wk) * 1024}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402   (hash of the kernel sources the passes ran on: bench.py reports traffic only when it matches)
json.dump({"note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch; separate --pmc passes of `bench.py --no-graph`",
           "kernel_src_sha": kernel_source_hash(), "kernels": res}, open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
