#!/bin/bash
# accumulation kernels alone (everything serialised on one stream, one proof at a time) vs pipelined with three in flight
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('ms/step', round(d['ms_per_step'],3), '| device total', round(t['total_ms'],2), 'acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'sum', round(t['acc_g1_ms']+t['acc_g2_ms'],3), 'plan', round(t['plan_ms'],2), 'poly', round(t['poly_ms'],2), 'reduce', round(t['reduce_ms'],2))"; }
for rep in 1 2; do
echo -n "serialised (GS_NO_OVERLAP=1, blocking): "; GS_NO_OVERLAP=1 run --steps 8 --warmup 2 --reps 3 --pipeline 1
echo -n "blocking, streams overlapped:           "; run --steps 8 --warmup 2 --reps 3 --pipeline 1
echo -n "three in flight:                        "; run --steps 10 --warmup 3 --reps 5
done
