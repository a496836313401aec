#!/bin/bash
# A/B of one library variant on the latency-sensitive cases: small blocking MSMs / proofs, plus the 2^20 lines
OUT=gpurun_out/${1:-ab5}; mkdir -p $OUT; V=$2
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_primitives.py tests/test_field_host.py -q -x 2>&1 | tail -3 ) | tee $OUT/pytest.txt
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'ms | acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'reduce', round(t['reduce_ms'],2))"; }
for round in 1 2; do
for v in "" $V; do
  if [ -n "$v" ]; then export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_$v.so; else unset GS_LIB; fi
  echo -n "msm 2^16 blocking, ${v:-default}: "; run --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3 --pipeline 1
  echo -n "msm 2^16 pipelined, ${v:-default}: "; run --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
  echo -n "msm 2^20 blocking, ${v:-default}: "; run --workload msm_g1 --steps 30 --warmup 3 --reps 3 --pipeline 1
  echo -n "prove 2^16 pipelined, ${v:-default}: "; run --log2n 16 --steps 100 --warmup 10 --reps 3
  echo -n "prove 2^16 blocking, ${v:-default}: "; run --log2n 16 --steps 60 --warmup 10 --reps 3 --pipeline 1
  echo -n "prove 2^20 pipelined, ${v:-default}: "; run --steps 10 --warmup 3 --reps 5
  echo -n "prove 2^20 blocking, ${v:-default}: "; run --steps 8 --warmup 2 --reps 3 --pipeline 1
done; done 2>&1 | tee $OUT/ab.txt
