#!/bin/bash
OUT=gpurun_out/${1:-blk}; mkdir -p $OUT
run() { echo -n "$1: "; shift; timeout 600 env "$@" python bench.py --steps 8 --warmup 2 --reps 3 --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'ms | acc g1', round(t['acc_g1_ms'],2), 'g2', round(t['acc_g2_ms'],2), 'plan', round(t['plan_ms'],2), 'poly', round(t['poly_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2; do
run "pipelined c17 filter" GS_PART_MIN_R=4
run "pipelined c17 partition" GS_PART_MIN_R=2
done

for v in 4 2; do echo -n "blocking part_min_r=$v: "; GS_PART_MIN_R=$v python bench.py --steps 8 --warmup 2 --reps 3 --cpu-log2n 0 --no-check --no-extras --pipeline 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'plan', round(t['plan_ms'],2))"; done
