#!/bin/bash
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('ms/step', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), '| acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'plan', round(t['plan_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2 3; do
for cfg in "32 4" "16 8"; do
  set -- $cfg; export GS_FOLD_MAX=$1 GS_REDUCE_L=$2
  echo -n "fold_max=$1 L_min=$2 | 2^18 pipelined: "; run --log2n 18 --steps 40 --warmup 5 --reps 3
  echo -n "fold_max=$1 L_min=$2 | 2^19 pipelined: "; run --log2n 19 --steps 20 --warmup 3 --reps 3
  echo -n "fold_max=$1 L_min=$2 | 2^17 pipelined: "; run --log2n 17 --steps 40 --warmup 5 --reps 3
done; done
