set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in 16 17 18; do bash tools/gpu_run.sh r4e ab tw1 r3tails : --log2n $L --steps 60 --warmup 10 --reps 3; done
bash tools/gpu_run.sh r4e ab tw1 r3tails : --log2n 16 --pipeline 1 --steps 40 --warmup 10 --reps 3
bash tools/gpu_run.sh r4e ab tw1 r3tails : --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
for a in "10" "14 --rows 16" "16 --rows 12" "18 --rows 8"; do python tools/derive_eval_basis.py --log2n $a 2>&1 | tail -1; done | tee gpurun_out/r4e/derive_eval.txt
