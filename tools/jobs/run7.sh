set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in 16 17 18 20; do bash tools/gpu_run.sh r4g ab alone r3tails : --log2n $L --steps $((L==20?10:60)) --warmup $((L==20?3:10)) --reps 3; done
bash tools/gpu_run.sh r4g ab alone r3tails : --workload msm_g1 --log2n 16 --pipeline 1 --steps 200 --warmup 20 --reps 3
