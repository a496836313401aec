set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DEV=$GRAFT_REPO_ROOT/gpurun_variants/lib_dev.so
bash tools/gpu_run.sh r4j env GS_LIB=$DEV GS_LIB=$DEV,GS_PLANW_STREAM=1 : --steps 10 --warmup 3 --reps 5
bash tools/gpu_run.sh r4j env GS_LIB=$DEV GS_LIB=$DEV,GS_PLANW_STREAM=1 : --workload prove_witness --steps 10 --warmup 3 --reps 3
bash tools/gpu_run.sh r4j env GS_LIB=$DEV GS_LIB=$DEV,GS_PLANW_STREAM=1 : --log2n 22 --steps 4 --warmup 1 --reps 3
GS_LIB=$DEV GS_PLANW_STREAM=1 bash tools/gpu_run.sh r4j trace planw_own --steps 4 --warmup 2 --reps 1 --cpu-log2n 0 --no-extras --no-check
