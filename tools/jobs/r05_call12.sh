# Round 5, twelfth GPU call: keys with sparse B arrays -- parity (the new test on both summation routes, the whole prover module),
# and the A/B against the single plan (GS_SPLIT_B_PERCENT=0) on the gates and the realistic instance.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5l
mkdir -p gpurun_out/$T
(timeout 1200 python -m pytest tests/test_gpu_prove.py tests/test_gpu_zy_multi.py -q --maxfail=5 2>&1 | tail -8) > gpurun_out/$T/pytest.txt; tail -4 gpurun_out/$T/pytest.txt
bash tools/gpu_run.sh $T env GS_SPLIT_B_PERCENT=0 : --instance gates --steps 10 --warmup 3 --reps 3
bash tools/gpu_run.sh $T env GS_SPLIT_B_PERCENT=0 : --instance gates --pipeline 1 --steps 8 --warmup 2 --reps 3
bash tools/gpu_run.sh $T env GS_SPLIT_B_PERCENT=0 : --instance realistic --steps 12 --warmup 3 --reps 3
bash tools/gpu_run.sh $T env GS_SPLIT_B_PERCENT=0 : --instance gates --workload prove_witness --steps 10 --warmup 3 --reps 3
