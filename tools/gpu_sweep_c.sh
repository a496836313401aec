#!/bin/bash
# window-width sweep on the 2^20 proof and the 2^20 / 2^16 G1 MSM (development probe)
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -q -x 2>&1 | tail -8 ) | tee $OUT/pytest_msm.txt
for c in 16 17 18 19 20; do
  echo -n "prove 2^20 c=$c: "
  timeout 600 python bench.py --steps 12 --warmup 3 --cpu-log2n 0 --no-check --window-bits $c 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print(round(d['ms_per_step'],3), 'ms | acc g1', round(t['acc_g1_ms'],2), 'g2', round(t['acc_g2_ms'],2), 'plan', round(t['plan_ms'],2), 'reduce', round(t['reduce_ms'],2), 'poly', round(t['poly_ms'],2))"
done 2>&1 | tee $OUT/sweep_prove.txt
for c in 16 18 19 20; do
  echo -n "prove 2^20 c=$c blocking: "
  timeout 600 python bench.py --steps 8 --warmup 2 --cpu-log2n 0 --no-check --window-bits $c --pipeline 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), 'ms')"
done 2>&1 | tee -a $OUT/sweep_prove.txt
for c in 16 19 20; do
  echo -n "msm_g1 2^20 c=$c: "
  timeout 600 python bench.py --workload msm_g1 --steps 20 --warmup 3 --cpu-log2n 0 --window-bits $c 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), 'ms')"
done 2>&1 | tee -a $OUT/sweep_prove.txt
