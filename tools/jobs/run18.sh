set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r4o env GS_PLANW_STREAM=2 GS_PLANW_STREAM=0 : --instance realistic --steps 10 --warmup 3 --reps 5
bash tools/gpu_run.sh r4o env GS_PLANW_STREAM=2 : --instance realistic --workload prove_witness --steps 10 --warmup 3 --reps 3
bash tools/gpu_run.sh r4o trace realistic --instance realistic --steps 6 --warmup 2 --reps 1 --cpu-log2n 0 --no-extras --no-check
