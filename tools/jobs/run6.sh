set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/gpurun_variants/lib_r3tails.so
bash tools/gpu_run.sh r4f env GS_TAIL_PRIORITY=1 GS_TAIL_PRIORITY=2 GS_LIB=$OLD
for L in 17 18; do bash tools/gpu_run.sh r4f env GS_TAIL_PRIORITY=1 GS_TAIL_PRIORITY=2 GS_LIB=$OLD : --log2n $L --steps 60 --warmup 10 --reps 3; done
