#!/bin/bash
# SQ instruction counters of BOTH accumulation kernels on the judged workload (one blocking 2^20 proof per step)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcg2}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for GRP in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $GRP --output-format csv -d $OUT/g$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --reps 1 --pipeline 1 --settle-ms 0 --cpu-log2n 0 --no-extras --no-check > /dev/null 2>&1
  F=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  for C in $GRP; do [ -n "$F" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$F" $C 40 | grep "k_bucket_accumulate" | sed "s/^/$C: /"; done
  rm -rf $OUT/g$i
done > $OUT/pmc_sq_accumulate_prove.txt 2>&1
cat $OUT/pmc_sq_accumulate_prove.txt
