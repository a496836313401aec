#!/bin/bash
OUT=gpurun_out/${1:-r3c}; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) | tee $OUT/pytest_gpu.txt
python bench.py --steps 10 --warmup 2 --reps 3 --cpu-log2n 0 2>&1 | tail -1 > $OUT/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/%s/bench.json" % "r3c"))
print({k: d[k] for k in ("ms_per_step","blocking_ms_per_proof","host_buffers_ms_per_step","from_r1cs_ms_per_step","from_r1cs_via_px_ms_per_step")})
print(d["msm_g1"])
PY
for t in 1 2 4 8; do echo -n "copy threads $t: "; GS_COPY_THREADS=$t python bench.py --steps 4 --warmup 1 --reps 1 --cpu-log2n 0 --no-check 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('host buffers', round(d['host_buffers_ms_per_step'],3), 'blocking', round(d['blocking_ms_per_proof'],3))"; done
