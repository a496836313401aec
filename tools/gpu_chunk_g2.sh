#!/bin/bash
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py -m gpu -q -x 2>&1 | tail -2 )
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('ms/step', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), '| acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2 3; do
for v in 1 0; do
  export GS_CHUNK_G2=$v
  echo -n "own G2 chunks=$v | 2^16 pipelined: "; run --log2n 16 --steps 100 --warmup 10 --reps 3
  echo -n "own G2 chunks=$v | 2^15 pipelined: "; run --log2n 15 --steps 100 --warmup 10 --reps 3
done; done
for v in 1 0; do
  export GS_CHUNK_G2=$v
  echo -n "own G2 chunks=$v | 2^14 pipelined: "; run --log2n 14 --steps 100 --warmup 10 --reps 3
  echo -n "own G2 chunks=$v | 2^16 blocking: "; run --log2n 16 --steps 60 --warmup 10 --reps 3 --pipeline 1
  echo -n "own G2 chunks=$v | 2^16 pinocchio: "; run --workload prove_pinocchio --log2n 16 --steps 60 --warmup 10 --reps 3
done
