#!/bin/bash
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('ms/step', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), '| acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'plan', round(t['plan_ms'],2), 'poly', round(t['poly_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2 3; do
echo -n "default (all aux streams high):   "; run --steps 10 --warmup 3 --reps 5
echo -n "tails low, plans/H(x) high:       "; GS_TAILS_LOW=1 run --steps 10 --warmup 3 --reps 5
echo -n "all aux at the main priority:     "; GS_NO_PRIORITY=1 run --steps 10 --warmup 3 --reps 5
done
echo -n "2^16 default:   "; run --log2n 16 --steps 100 --warmup 10 --reps 3
echo -n "2^16 tails low: "; GS_TAILS_LOW=1 run --log2n 16 --steps 100 --warmup 10 --reps 3
echo -n "blocking default:   "; run --steps 8 --warmup 2 --reps 3 --pipeline 1
echo -n "blocking tails low: "; GS_TAILS_LOW=1 run --steps 8 --warmup 2 --reps 3 --pipeline 1
