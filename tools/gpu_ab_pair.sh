#!/bin/bash
# A/B of library variants (GS_LIB) on the 2^20 proof: tests first (default library), then alternating bench runs
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py -m gpu -q -x 2>&1 | tail -5 ) | tee $OUT/pytest.txt
for round in 1 2; do
for v in "" nopair; do
  if [ -n "$v" ]; then export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_$v.so; else unset GS_LIB; fi
  echo -n "variant ${v:-default}: "
  python bench.py --steps 10 --warmup 3 --reps 5 --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'ms | acc g1', round(t['acc_g1_ms'],2), 'g2', round(t['acc_g2_ms'],2), 'c', d['config']['window_bits'])"
done; done 2>&1 | tee $OUT/ab.txt
unset GS_LIB
for v in "" nopair; do
  if [ -n "$v" ]; then export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_$v.so; else unset GS_LIB; fi
  echo -n "blocking, variant ${v:-default}: "
  python bench.py --steps 6 --warmup 2 --reps 3 --cpu-log2n 0 --no-check --no-extras --pipeline 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'ms | acc g1', round(t['acc_g1_ms'],2), 'g2', round(t['acc_g2_ms'],2))"
done 2>&1 | tee -a $OUT/ab.txt
