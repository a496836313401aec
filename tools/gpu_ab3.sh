#!/bin/bash
# A/B of several library variants on the pipelined 2^20 proof and the 2^16 proof, interleaved, three rounds
OUT=gpurun_out/${1:-ab3}; mkdir -p $OUT; shift
run() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['device_ms_per_step']; print('median', round(d['ms_per_step'],3), 'min', round(d['ms_per_step_min'],3), 'ms | acc g1', round(t['acc_g1_ms'],3), 'g2', round(t['acc_g2_ms'],3), 'reduce', round(t['reduce_ms'],2))"; }
for round in 1 2 3; do
for v in "" "$@"; do
  if [ -n "$v" ]; then export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_$v.so; else unset GS_LIB; fi
  echo -n "2^20 pipelined, ${v:-default}: "; run --steps 10 --warmup 3 --reps 5
done; done 2>&1 | tee $OUT/ab.txt
for v in "" "$@"; do
  if [ -n "$v" ]; then export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_$v.so; else unset GS_LIB; fi
  echo -n "2^16 pipelined, ${v:-default}: "; run --log2n 16 --steps 100 --warmup 10 --reps 3
  echo -n "2^20 blocking, ${v:-default}: "; run --steps 8 --warmup 2 --reps 3 --pipeline 1
done 2>&1 | tee -a $OUT/ab.txt
