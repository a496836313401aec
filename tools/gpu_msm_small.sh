#!/bin/bash
run() { echo -n "$1: "; shift; env "$@" python bench.py --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3 --cpu-log2n 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],4), 'ms', 'c', d['config']['window_bits'], {k: round(v,3) for k,v in t.items()})"; }
run "2^16 pipelined default" A=1
run "2^16 pipelined plan on aux1" GS_PLANW_STREAM=1
run "2^16 pipelined c=16" GS_BENCH_C=16
run "2^16 pipelined c=16 plan aux1" GS_BENCH_C=16 GS_PLANW_STREAM=1
run "2^16 pipelined c=14" GS_BENCH_C=14
