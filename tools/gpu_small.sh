#!/bin/bash
run() { echo -n "$1: "; shift; python bench.py "$@" --cpu-log2n 0 --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('median', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'M/s c', d['config']['window_bits'])"; }
run "prove 2^16 pipelined" --log2n 16 --steps 100 --warmup 10 --reps 3 --no-check
run "prove 2^18 pipelined" --log2n 18 --steps 40 --warmup 5 --reps 3 --no-check
run "prove 2^20 pipelined" --steps 10 --warmup 3 --reps 5 --no-check
run "msm 2^16 pipelined" --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
run "msm 2^20 pipelined" --workload msm_g1 --steps 40 --warmup 5 --reps 3
run "prove 2^22 c=auto" --log2n 22 --steps 5 --warmup 2 --reps 3 --no-check
run "prove 2^22 c=17" --log2n 22 --steps 5 --warmup 2 --reps 3 --no-check --window-bits 17
run "prove 2^22 c=18" --log2n 22 --steps 5 --warmup 2 --reps 3 --no-check --window-bits 18
