#!/bin/bash
# usage (GPU box, repo root): BENCH_ARGS="--nodes 512 --hidden 128 --steps 50 --warmup 5" tools/ab_env_args.sh <rounds> "VAR=a" "VAR=b" ...
# as tools/ab_env.sh for another bench configuration
rounds=$1; shift
for r in $(seq $rounds); do
  for e in "$@"; do
    v=$(env $e python bench.py $BENCH_ARGS --no-cpu-baseline --no-kernel-timing --no-module-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f steps/s  %.4f ms  random-mask %s' % (d['value'], d['ms_per_step'], d.get('steps_per_s_random_mask_phase')))")
    echo "[${e:-default}]: $v"
  done
done
