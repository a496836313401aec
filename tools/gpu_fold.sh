#!/bin/bash
( timeout 600 python -m pytest tests/test_gpu_msm.py -m gpu -q -x 2>&1 | tail -2 )
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('ms/step', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), '| acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'plan', round(t['plan_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2; do
for cfg in "16 8" "32 8" "64 4" "32 4" "64 2"; do
  set -- $cfg; export GS_FOLD_MAX=$1 GS_REDUCE_L=$2
  echo -n "fold_max=$1 L_min=$2 | 2^20 pipelined: "; run --steps 10 --warmup 3 --reps 5
done; done
for cfg in "16 8" "32 8" "64 4" "64 2"; do
  set -- $cfg; export GS_FOLD_MAX=$1 GS_REDUCE_L=$2
  echo -n "fold_max=$1 L_min=$2 | 2^20 blocking: "; run --steps 8 --warmup 2 --reps 3 --pipeline 1
  echo -n "fold_max=$1 L_min=$2 | 2^16 pipelined: "; run --log2n 16 --steps 100 --warmup 10 --reps 3
  echo -n "fold_max=$1 L_min=$2 | msm 2^16 blocking: "; run --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3 --pipeline 1
  echo -n "fold_max=$1 L_min=$2 | serialised 2^20: "; GS_NO_OVERLAP=1 run --steps 6 --warmup 2 --reps 1 --pipeline 1
done
