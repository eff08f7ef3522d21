#!/bin/bash
# usage (GPU box, repo root): tools/results_table.sh > gpurun_out/table.txt  -- the rows of BASELINE.md section 7 from one box
cd $GRAFT_REPO_ROOT
b() { python bench.py --no-cpu-baseline --no-kernel-timing --no-module-path "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%8.1f steps/s  %.3f ms  rnd %s' % (d['value'], d['ms_per_step'], d.get('steps_per_s_random_mask_phase')))"; }
echo "c2 default:      $(b --steps 200 --warmup 20)"
echo "c2 deterministic: $(GPTST_DETERMINISTIC=1 b --steps 200 --warmup 20)"
echo "c2 shard nodes w1: $(b --steps 200 --warmup 20 --shard nodes)"
echo "c3 METR_LA:      $(b --steps 200 --warmup 20 --dataset METR_LA)"
for hs in 2 5 10 20 40; do echo "c4 NYC_TAXI HS=$hs: $(b --steps 100 --warmup 10 --dataset NYC_TAXI --hs $hs)"; done
echo "c5 N4096 C128 B32: $(b --steps 10 --warmup 3 --nodes 4096 --hidden 128)"
echo "c5 N4096 C128 B8:  $(b --steps 20 --warmup 3 --nodes 4096 --hidden 128 --batch 8)"
echo "c5 shard share N512 C128: $(b --steps 50 --warmup 5 --nodes 512 --hidden 128)"
echo "c5 shard share N512 C128 sharded w1: $(b --steps 50 --warmup 5 --nodes 512 --hidden 128 --shard nodes)"
echo "c2 FORCE_DP native: $(GPTST_FORCE_DP=1 b --steps 200 --warmup 20)"
echo "c2 FORCE_DP torch:  $(GPTST_FORCE_DP=1 b --steps 200 --warmup 20 --torch-comm)"
echo "c2 B=16: $(b --steps 200 --warmup 20 --batch 16)   B=64: $(b --steps 100 --warmup 10 --batch 64)"
