#!/bin/bash
OUT=gpurun_out/${1:-ab2}; mkdir -p $OUT; V=$2
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_primitives.py -m gpu -q -x 2>&1 | tail -3 ) | tee $OUT/pytest.txt
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'ms | acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'reduce', round(t['reduce_ms'],2))"; }
for round in 1 2 3; do
for v in "" $V; do
  if [ -n "$v" ]; then export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_$v.so; else unset GS_LIB; fi
  echo -n "2^20 pipelined, ${v:-default}: "; run --steps 10 --warmup 3 --reps 5
  echo -n "msm 2^20 blocking, ${v:-default}: "; run --workload msm_g1 --steps 20 --warmup 3 --reps 3 --pipeline 1
done; done 2>&1 | tee $OUT/ab.txt
