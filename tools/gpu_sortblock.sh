#!/bin/bash
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'ms | total', round(t['total_ms'],2), 'acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'plan', round(t['plan_ms'],2), 'poly', round(t['poly_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2; do
for sb in 1024 512 256; do
  export GS_SORT_BLOCK=$sb
  echo -n "2^20 prove, sort block $sb: "; run --steps 10 --warmup 3 --reps 5
  echo -n "2^20 msm,   sort block $sb: "; run --workload msm_g1 --steps 40 --warmup 5 --reps 3
done; done
for sb in 1024 512 256; do
  export GS_SORT_BLOCK=$sb
  echo -n "2^20 blocking, sort block $sb: "; run --steps 8 --warmup 2 --reps 3 --pipeline 1
  echo -n "2^16 prove, sort block $sb: "; run --log2n 16 --steps 100 --warmup 10 --reps 3
done
