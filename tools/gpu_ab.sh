#!/bin/bash
for v in "--pipeline 2" "--pipeline 2" "--pipeline 1"; do echo "== $v"; python bench.py --steps 12 --warmup 2 --cpu-log2n 0 --no-check $v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value']/1e6,'Mc/s', d['ms_per_step'], d['device_ms_per_step'])"; done
