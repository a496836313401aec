#!/bin/bash
# One gpurun call of the dev loop: all GPU parity tests (no -x: every failure in one trip), smoke, a short bench line.
# usage: tools/gpu_check.sh <tag> [bench args...]
TAG=${1:-check}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $OUT/pytest_gpu.txt
tail -40 $OUT/pytest_gpu.txt
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) | tee $OUT/smoke.txt
( timeout 900 python bench.py --steps 10 --warmup 2 --cpu-log2n 0 "$@" 2>&1 | tail -3 ) | tee $OUT/bench.txt
