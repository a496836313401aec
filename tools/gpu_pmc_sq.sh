#!/bin/bash
# VALU-issue evidence for the dominant kernel: SQ counters on the 2^20 G1 MSM workload (one pass per counter group)
TAG=${1:-pmcsq}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for GRP in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $GRP --output-format csv -d $OUT/g$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload msm_g1 --steps 2 --warmup 1 --cpu-log2n 0 > $OUT/run_g$i.txt 2>&1
  F=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  for C in $GRP; do [ -n "$F" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$F" $C | grep -i "kernel,\|k_bucket_accumulate" | tail -1 | sed "s/^/$C: /"; done
  rm -rf $OUT/g$i
done 2>&1 | tee $OUT/sq_summary.txt
