set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DEV=$GRAFT_REPO_ROOT/gpurun_variants/lib_dev.so
bash tools/gpu_run.sh r4h tests
bash tools/gpu_run.sh r4h env GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=0 GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=32 : --log2n 19 --steps 20 --warmup 5 --reps 3
bash tools/gpu_run.sh r4h env GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=0 GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=32 : --log2n 19 --pipeline 1 --steps 20 --warmup 5 --reps 3
bash tools/gpu_run.sh r4h env GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=0 GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=32 : --log2n 20 --pipeline 1 --steps 10 --warmup 3 --reps 3
bash tools/gpu_run.sh r4h env GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=0 GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=32 : --workload msm_g1 --log2n 20 --steps 40 --warmup 5 --reps 3
bash tools/gpu_run.sh r4h env GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=0 GS_LIB=$DEV,GS_TAIL_ALONE_LOG2=32 : --workload msm_g1 --log2n 20 --pipeline 1 --steps 40 --warmup 5 --reps 3
