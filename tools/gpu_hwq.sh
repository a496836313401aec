#!/bin/bash
run() { echo -n "$1: "; shift; env "$@" python bench.py --steps 10 --warmup 3 --reps 5 --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'ms | acc g1', round(t['acc_g1_ms'],2), 'g2', round(t['acc_g2_ms'],2), 'plan', round(t['plan_ms'],2), 'poly', round(t['poly_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
runm() { echo -n "$1: "; shift; env "$@" python bench.py --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3 --cpu-log2n 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],4), 'ms plan', round(t['plan_ms'],3))"; }
GS_DEBUG_STREAMS=1 python -c "
import gosnark_amd; from gosnark_amd import capi; capi.init(0)" 2>&1 | grep gosnark
for p in 99 0 1; do
run "prove aux3prio=$p" GS_AUX3_PRIO=$p
runm "msm 2^16 aux3prio=$p" GS_AUX3_PRIO=$p
done
run "prove planw on aux1" GS_PLANW_STREAM=1
runm "msm 2^20 default" A=1 
