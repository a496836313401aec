set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s
D=/tmp/prof_msm; rm -rf $D
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $GRAFT_REPO_ROOT/bench.py --workload msm_g1 --steps 30 --warmup 5 --reps 1 --cpu-log2n 0 --no-extras --no-check > $GRAFT_REPO_ROOT/gpurun_out/r4s/run.txt 2>&1 )
T=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/trace_steady.py $T "k_bucket_accumulate<gs::FqTag>" 25 3 > gpurun_out/r4s/timeline_msm_g1_steady.txt; head -70 gpurun_out/r4s/timeline_msm_g1_steady.txt
D=/tmp/prof_prove; rm -rf $D
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --reps 1 --cpu-log2n 0 --no-extras --no-check > $GRAFT_REPO_ROOT/gpurun_out/r4s/run2.txt 2>&1 )
T=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/trace_steady.py $T "k_bucket_accumulate<gs::Fq2Tag>" 8 2 > gpurun_out/r4s/timeline_prove_steady.txt; tail -30 gpurun_out/r4s/timeline_prove_steady.txt
