#!/bin/bash
for l in 20; do for ch in 16 32 64 128; do echo -n "log2n $l chunk $ch: "; GS_CHUNK=$ch python bench.py --log2n $l --steps 16 --warmup 3 --cpu-log2n 0 --no-check 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],3), 'ms')"; done; done
for ch in 16 32 64; do echo -n "msm 2^20 chunk $ch: "; GS_CHUNK=$ch python bench.py --workload msm_g1 --log2n 20 --steps 30 --warmup 4 --cpu-log2n 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],3), 'ms')"; done
