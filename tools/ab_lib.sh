#!/bin/bash
# usage (GPU box, repo root): tools/ab_lib.sh <rounds> libA.so libB.so ...   -- bench.py with each build of the library in turn, on ONE box
# (box-to-box spread is ~1 %, larger than most kernel-level changes).  Build the variants with `python -m gptst_amd.build` and copy
# gpt-st_amd/lib/libgptst_hip.so aside under another name in gpt-st_amd/lib/ (built .so files travel with the snapshot).
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    v=$(GPTST_LIB=$PWD/$lib python bench.py --steps 320 --warmup 24 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.4f ms  random-mask %.1f' % (d['value'], d['ms_per_step'], d['steps_per_s_random_mask_phase']))")
    echo "$lib: $v"
  done
done
