#!/bin/bash
( timeout 600 python -m pytest tests/test_gpu_msm.py -m gpu -q -x 2>&1 | tail -1 )
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('ms/step', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), '| acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'plan', round(t['plan_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2; do
for L in 16 17 18 19 20; do echo -n "2^$L pipelined: "; run --log2n $L --steps $((L<18?60:(L<20?20:10))) --warmup 5 --reps 3; done
done
echo -n "msm 2^16 blocking: "; run --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3 --pipeline 1
echo -n "msm 2^16 pipelined: "; run --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
