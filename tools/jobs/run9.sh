set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in 14 16 17 18; do bash tools/gpu_run.sh r4i ab acc1 acc2 : --log2n $L --steps 60 --warmup 10 --reps 3; done
bash tools/gpu_run.sh r4i ab acc1 acc2 : --workload msm_g1 --log2n 16 --pipeline 1 --steps 200 --warmup 20 --reps 3
bash tools/gpu_run.sh r4i ab acc1 acc2 : --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
bash tools/gpu_run.sh r4i ab acc1 acc2 : --log2n 16 --pipeline 1 --steps 40 --warmup 10 --reps 3
