#!/bin/bash
# Round-2 evidence in ONE gpurun call -> gpurun_out/<tag>/ (copied into profiles/ afterwards):
#   the judged bench line, rocprofv3 kernel stats of the same command, PMC passes (HBM traffic; SQ instruction counters),
#   kernel timelines (pipelined / blocking), the other sizes and workloads, the logical-shard stand-ins of configs[3].
TAG=${1:-r02prof}; COMMIT=${2:-unknown}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps(d, indent=1))"; }
( timeout 900 python bench.py 2>$OUT/bench_stderr.txt | line ) > $OUT/bench_line.json
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print('judged line:', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M constraints/s', 'blocking', round(d.get('blocking_ms_per_proof',0),2), 'from_r1cs', round(d.get('from_r1cs_ms_per_step',0),2))"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --reps 1 --cpu-log2n 0 --no-extras > $OUT/bench_under_rocprof.txt 2>&1
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $OUT/rocprofv3_kernel_stats_bench_steps3.csv
F=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$F" ] && python $GRAFT_REPO_ROOT/tools/trace_window.py "$F" 1 2 > $OUT/timeline_pipelined.txt 2>&1
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --reps 1 --pipeline 1 --cpu-log2n 0 --no-extras --no-check > /dev/null 2>&1
F=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$F" ] && python $GRAFT_REPO_ROOT/tools/trace_window.py "$F" 2 1 > $OUT/timeline_blocking.txt 2>&1
rm -rf $OUT/prof
# --- PMC: HBM traffic of the accumulate kernels on the judged workload (separate passes, kernel names only)
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $CTR --output-format csv -d $OUT/$CTR -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --reps 1 --cpu-log2n 0 --no-extras --no-check > /dev/null 2>&1
  F=$(find $OUT/$CTR -name "*counter_collection.csv" | head -1)
  echo "== $CTR"; [ -n "$F" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$F" $CTR 20
  rm -rf $OUT/$CTR
done > $OUT/pmc_fetch_write_size_prove_2p20.txt 2>&1
# --- PMC: SQ counters of a lone 2^20-term G1 accumulation
i=0
for GRP in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $GRP --output-format csv -d $OUT/g$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload msm_g1 --steps 2 --warmup 1 --reps 1 --pipeline 1 --cpu-log2n 0 > /dev/null 2>&1
  F=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  for C in $GRP; do [ -n "$F" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$F" $C 30 | grep "k_bucket_accumulate" | tail -1 | sed "s/^/$C: /"; done
  rm -rf $OUT/g$i
done > $OUT/pmc_sq_accumulate_g1.txt 2>&1
cd $GRAFT_REPO_ROOT
# --- the other sizes / workloads (median of 3 repetitions each, checks on)
for L in 16 18 22; do ( timeout 900 python bench.py --log2n $L --reps 3 --cpu-log2n 0 --no-extras 2>/dev/null | line ) > $OUT/bench_line_2p$L.json; done
( timeout 600 python bench.py --workload msm_g1 --steps 40 --reps 3 --cpu-log2n 0 2>/dev/null | line ) > $OUT/bench_line_msm_g1.json
( timeout 600 python bench.py --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3 --cpu-log2n 0 2>/dev/null | line ) > $OUT/bench_line_msm_g1_2p16.json
( timeout 600 python bench.py --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3 --pipeline 1 --cpu-log2n 0 2>/dev/null | line ) > $OUT/bench_line_msm_g1_2p16_blocking.json
( timeout 600 python bench.py --workload prove_pinocchio --reps 3 --cpu-log2n 0 2>/dev/null | line ) > $OUT/bench_line_pinocchio.json
( timeout 900 python bench.py --workload prove_sharded --logical-shards 8 --log2n 22 --steps 3 --warmup 1 --reps 3 --cpu-log2n 0 2>$OUT/sharded_stderr.txt | line ) > $OUT/bench_line_prove_sharded_8_logical_2p22.json
( timeout 900 python bench.py --workload msm_sharded --logical-shards 8 --log2n 22 --steps 5 --warmup 1 --reps 3 --cpu-log2n 0 2>>$OUT/sharded_stderr.txt | line ) > $OUT/bench_line_msm_sharded_8_logical_2p22.json
( timeout 900 python bench.py --workload prove_sharded --logical-shards 8 --log2n 20 --steps 5 --warmup 1 --reps 3 --cpu-log2n 0 2>>$OUT/sharded_stderr.txt | line ) > $OUT/bench_line_prove_sharded_8_logical_2p20.json
ls -la $OUT | head -40
# --- the multi-process code path of bench.py on a 1-GPU box (two ranks share GPU 0, gloo): barriers, max-over-ranks, one line
( GS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --log2n 16 --steps 20 --warmup 3 --reps 2 --cpu-log2n 0 2>$OUT/two_ranks_stderr.txt | line ) > $OUT/bench_line_two_ranks_sharing_one_gpu_2p16.json
ls $OUT | wc -l
