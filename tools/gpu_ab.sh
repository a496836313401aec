#!/bin/bash
# A/B the aux-stream CU mask on the 2^20 prove
for s in 0 8 16 4; do echo "== GS_AUX_CU_STRIDE=$s"; GS_AUX_CU_STRIDE=$s python bench.py --steps 5 --warmup 2 --cpu-log2n 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value']/1e6,'Mc/s', d['ms_per_step'], d['device_ms_per_step'])"; done
