#!/bin/bash
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'ms | acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2; do
for ch in 16 12 20 24; do
  export GS_CHUNK=$ch
  echo -n "2^16 chunk $ch: "; run --log2n 16 --steps 100 --warmup 10 --reps 3
done; done
for ch in 16 20 24 32; do
  export GS_CHUNK=$ch
  echo -n "2^17 chunk $ch: "; run --log2n 17 --steps 60 --warmup 10 --reps 3
  echo -n "msm 2^16 chunk $ch: "; run --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
done
