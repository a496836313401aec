#!/bin/bash
# kernel timeline of a short bench run (rocprofv3 --kernel-trace, csv) -> gpurun_out/<tag>/
TAG=${1:-trace}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-log2n 0 "$@" > $OUT/bench_under_rocprof.txt 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp "$S" $OUT/kernel_stats.csv
python tools/trace_timeline.py "$F" > $OUT/timeline.txt 2>&1
rm -rf $OUT/prof
tail -60 $OUT/timeline.txt
