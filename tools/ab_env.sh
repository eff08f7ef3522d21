#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh <rounds> "VAR=a" "VAR=b" ...   -- bench.py under each environment setting in turn, on ONE box
# (box-to-box spread is ~1 %).  An empty string "" is the default environment.
rounds=$1; shift
for r in $(seq $rounds); do
  for e in "$@"; do
    v=$(env $e python bench.py --steps 320 --warmup 24 --no-cpu-baseline --no-kernel-timing --no-module-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.4f ms  random-mask %.1f  timeouts %s' % (d['value'], d['ms_per_step'], d['steps_per_s_random_mask_phase'], d.get('handoff_timeouts')))")
    echo "[${e:-default}]: $v"
  done
done
