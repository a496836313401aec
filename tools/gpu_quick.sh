#!/bin/bash
# quick dev loop on the GPU box: parity tests + MSM timing probe + short bench
TAG=${1:-quick}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) | tee $OUT/pytest_gpu.txt
( timeout 600 python tools/quick_msm_bench.py 2>&1 | tail -20 ) | tee $OUT/quick_msm.txt
( timeout 600 python bench.py --steps 5 --warmup 1 --cpu-log2n 0 "$@" 2>&1 | tail -3 ) | tee $OUT/bench.txt
