#!/bin/bash
OUT=gpurun_out/${1:-r3}; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $OUT/pytest_gpu.txt
tail -25 $OUT/pytest_gpu.txt
blk() { echo -n "blocking $1: "; shift; env "$@" python bench.py --steps 10 --warmup 2 --reps 5 --cpu-log2n 0 --no-check --no-extras --pipeline 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'dev total', round(t['total_ms'],2), 'acc g1', round(t['acc_g1_ms'],2), 'g2', round(t['acc_g2_ms'],2))"; }
for rep in 1 2; do
blk "early_poly=1" GS_EARLY_POLY=1
blk "early_poly=0" GS_EARLY_POLY=0
done 2>&1 | tee $OUT/blocking_ab.txt
GS_HOST_TRACE=1 python bench.py --steps 4 --warmup 2 --reps 1 --cpu-log2n 0 --no-check --no-extras --pipeline 1 2>&1 | grep "gs host" | tail -6 | tee $OUT/host_trace.txt
python bench.py --steps 10 --warmup 2 --reps 3 --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | cut -c1-330
python bench.py --workload prove_pinocchio --steps 10 --warmup 2 --reps 3 --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | cut -c1-330
