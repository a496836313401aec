#!/bin/bash
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'ms | acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'plan', round(t['plan_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2 3; do
for ch in 16 24 32 20; do
  export GS_CHUNK=$ch
  echo -n "2^20 chunk $ch: "; run --steps 10 --warmup 3 --reps 5
done; done
for ch in 16 24 32; do
  export GS_CHUNK=$ch
  echo -n "2^20 blocking chunk $ch: "; run --steps 8 --warmup 2 --reps 3 --pipeline 1
  echo -n "msm 2^20 pipelined chunk $ch: "; run --workload msm_g1 --steps 40 --warmup 5 --reps 3
done
