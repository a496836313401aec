#!/bin/bash
OUT=gpurun_out/${1:-exp1}; mkdir -p $OUT
run() { echo -n "$1: "; shift; timeout 600 env "$@" python bench.py --steps 12 --warmup 3 --cpu-log2n 0 --no-check 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print(round(d['ms_per_step'],3), 'ms | acc g1', round(t['acc_g1_ms'],2), 'g2', round(t['acc_g2_ms'],2), 'plan', round(t['plan_ms'],2), 'reduce', round(t['reduce_ms'],2), 'poly', round(t['poly_ms'],2))"; }
for c in 16 17 20; do
 for sb in 1024 512 256; do
  run "c=$c sortblock=$sb planw=aux1" GS_SORT_BLOCK=$sb GS_BENCH_C=$c
  run "c=$c sortblock=$sb planw=aux0" GS_SORT_BLOCK=$sb GS_BENCH_C=$c GS_PLANW_STREAM=0
 done
done 2>&1 | tee $OUT/exp1.txt
