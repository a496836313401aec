set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in 16 17 18; do bash tools/gpu_run.sh r4l env GS_PLANW_STREAM=2 : --workload prove_witness --log2n $L --steps 60 --warmup 10 --reps 3; done
for L in 16 18; do bash tools/gpu_run.sh r4l env GS_PLANW_STREAM=2 : --log2n $L --steps 60 --warmup 10 --reps 3; done
