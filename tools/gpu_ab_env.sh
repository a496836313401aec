#!/bin/bash
# alternating A/B of runtime switches on the judged workload (each line: median / min of 5 repetitions of 10 proofs)
run() { echo -n "$1: "; shift; env "$@" python bench.py --steps 10 --warmup 3 --reps 5 --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'ms | acc g1', round(t['acc_g1_ms'],2), 'g2', round(t['acc_g2_ms'],2), 'plan', round(t['plan_ms'],2), 'poly', round(t['poly_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for round in 1 2 3; do
run "L<=32 heavy64 " A=1
run "L<=8  heavy64 " GS_REDUCE_MAXL=8
run "L<=32 heavy1024" GS_HEAVY_GRID=1024
run "L<=8  heavy1024" GS_REDUCE_MAXL=8 GS_HEAVY_GRID=1024
done
