#!/bin/bash
run() { echo -n "$1: "; shift; python bench.py "$@" --cpu-log2n 0 --no-extras --no-check 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('median', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'M/s c', d['config']['window_bits'])"; }
for rep in 1 2; do
for f in 0 1; do
export GS_TAIL_FLIP=$f
run "flip=$f prove 2^20" --steps 10 --warmup 3 --reps 5
run "flip=$f prove 2^19" --log2n 19 --steps 20 --warmup 3 --reps 3
run "flip=$f prove 2^18" --log2n 18 --steps 40 --warmup 5 --reps 3
run "flip=$f prove 2^17" --log2n 17 --steps 40 --warmup 5 --reps 3
done; done
