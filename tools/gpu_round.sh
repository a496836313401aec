#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench line, rocprofv3 kernel stats of the same bench command.
# usage: tools/gpu_round.sh <tag> [bench args...]
TAG=${1:-r01}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > $OUT/smoke.txt
cat $OUT/smoke.txt
( timeout 900 python bench.py "$@" 2>&1 | tail -5 ) > $OUT/bench.txt
cat $OUT/bench.txt
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-log2n 0 2>&1 | tail -3 )
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -30 "$F" > $OUT/kernel_stats_head.csv
find $OUT/prof -name "*.db" -delete 2>/dev/null
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
