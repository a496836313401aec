#!/bin/bash
OUT=gpurun_out/${1:-r3}; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_c_drivers.py tests/test_gpu_prove.py -m gpu -q 2>&1 | tail -30 ) > $OUT/pytest_gpu.txt
tail -12 $OUT/pytest_gpu.txt
GS_HOST_TRACE=1 python bench.py --steps 4 --warmup 2 --reps 1 --cpu-log2n 0 --no-check --no-extras --pipeline 1 2>&1 | grep "gs host" | tail -6 | tee $OUT/host_trace.txt
for rep in 1 2; do
python bench.py --steps 10 --warmup 2 --reps 5 --cpu-log2n 0 --no-check --no-extras --pipeline 1 2>&1 | tail -1 | cut -c1-330
python bench.py --steps 10 --warmup 2 --reps 5 --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | cut -c1-330
done
python bench.py --log2n 16 --steps 100 --warmup 10 --reps 3 --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 | cut -c1-330
