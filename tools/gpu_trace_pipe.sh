#!/bin/bash
# kernel timeline of the LAST pipelined steps (three proofs in flight) -> gpurun_out/<tag>/timeline.txt
TAG=${1:-trace}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-log2n 0 --no-check "$@" > $OUT/bench_under_rocprof.txt 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp "$S" $OUT/kernel_stats.csv
python tools/trace_window.py "$F" 4 2 > $OUT/timeline.txt
rm -rf $OUT/prof
