#!/bin/bash
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'ms | acc g1', round(t['acc_g1_ms'],3), 'plan', round(t['plan_ms'],3), 'reduce', round(t['reduce_ms'],2))"; }
for i in 1 2 3; do
echo -n "msm 2^16 pipelined: "; run --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
echo -n "msm 2^16 pipelined, no flip: "; GS_TAIL_FLIP=0 run --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
done
echo -n "msm 2^16 blocking: "; run --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3 --pipeline 1
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
import bench
import gosnark_amd
from gosnark_amd import capi
capi.init(0)
for i in range(3):
    print(bench.msm_extras(0x5EED0002)["2^16"])
PY
