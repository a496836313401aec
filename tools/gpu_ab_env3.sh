#!/bin/bash
OUT=gpurun_out/${1:-abe}; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_zy_multi.py -m gpu -q -x 2>&1 | tail -3 ) | tee $OUT/pytest.txt
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'ms | total', round(t['total_ms'],2), 'acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'plan', round(t['plan_ms'],2), 'poly', round(t['poly_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for round in 1 2 3; do
for v in 0 1; do
  export GS_G1_FIRST=$v
  echo -n "2^20 pipelined, g1_first=$v: "; run --steps 10 --warmup 3 --reps 5
done; done 2>&1 | tee $OUT/ab.txt
for v in 0 1; do
  export GS_G1_FIRST=$v
  echo -n "2^16 pipelined, g1_first=$v: "; run --log2n 16 --steps 100 --warmup 10 --reps 3
  echo -n "2^18 pipelined, g1_first=$v: "; run --log2n 18 --steps 40 --warmup 5 --reps 3
  echo -n "2^22 pipelined, g1_first=$v: "; run --log2n 22 --steps 5 --warmup 2 --reps 3
done 2>&1 | tee -a $OUT/ab.txt
