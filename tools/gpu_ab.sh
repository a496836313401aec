#!/bin/bash
for v in "" "GS_NO_PRIORITY=1" "" "GS_NO_PRIORITY=1"; do echo "== $v"; env $v python bench.py --steps 8 --warmup 2 --cpu-log2n 0 --no-check --instance random 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value']/1e6,'Mc/s', d['ms_per_step'], d['device_ms_per_step'])"; done
