#!/bin/bash
# HBM traffic of the dominant kernel from the PMC counters (separate passes, no tracing domains besides kernel-trace)
TAG=${1:-pmc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $CTR --output-format csv -d $OUT/$CTR -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-log2n 0 "$@" > $OUT/${CTR}_run.txt 2>&1
  F=$(find $OUT/$CTR -name "*counter_collection.csv" | head -1)
  echo "== $CTR ($F)"
  [ -n "$F" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$F" $CTR | tee $OUT/${CTR}_summary.txt
  rm -rf $OUT/$CTR
done
