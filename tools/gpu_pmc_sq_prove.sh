#!/bin/bash
# SQ counters for the main kernels of one Groth16 proof (blocking calls, so launches do not overlap): one rocprofv3 --pmc pass
# per counter group (no tracing domains), per-kernel averages -> gpurun_out/<tag>/sq_prove_summary.txt
TAG=${1:-pmcsq_prove}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
KERNELS='k_bucket_accumulate<gs::FqTag>|k_bucket_accumulate<gs::Fq2Tag>|k_ntt_pass<false>|k_ntt_pass<true>|k_scatter|k_hist|k_block_reduce<gs::FqTag>|k_bucket_combine<gs::FqTag>|k_heavy_combine<gs::FqTag>'
i=0
for GRP in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $GRP --output-format csv -d $OUT/g$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --pipeline 1 --steps 2 --warmup 1 --cpu-log2n 0 --no-check > $OUT/run_g$i.txt 2>&1
  F=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  for C in $GRP; do [ -n "$F" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$F" $C 40 | grep -E "$KERNELS" | sed "s/^/$C: /"; done
  rm -rf $OUT/g$i
done 2>&1 | tee $OUT/sq_prove_summary.txt
