#!/bin/bash
# A/B of library variants (GS_LIB) on the 2^20 prove: base, variant, base, variant
for v in "" "$1" "" "$1"; do
  if [ -n "$v" ]; then export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_$v.so; else unset GS_LIB; fi
  echo -n "variant ${v:-base}: "
  python bench.py --steps 16 --warmup 3 --cpu-log2n 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],3), 'ms', 'g1', round(d['device_ms_per_step']['acc_g1_ms'],2), 'g2', round(d['device_ms_per_step']['acc_g2_ms'],2), d.get('proof_verified','')[:24])"
done
