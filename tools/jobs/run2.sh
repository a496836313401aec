set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DEV=$GRAFT_REPO_ROOT/gpurun_variants/lib_dev.so
bash tools/gpu_run.sh r4b tests
bash tools/scale_selftest.sh 2 > gpurun_out/r4b/selftest.txt 2>&1; tail -12 gpurun_out/r4b/selftest.txt
(python bench.py --instance realistic --cpu-log2n 0 --reps 3 2>&1 | tail -1) > gpurun_out/r4b/bench_realistic.json; head -c 600 gpurun_out/r4b/bench_realistic.json
bash tools/gpu_run.sh r4b env GS_LIB=$DEV GS_LIB=$DEV,GS_ACC_STREAMS=2
