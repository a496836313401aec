set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r4r ab s9 : --steps 10 --warmup 3 --reps 5
bash tools/gpu_run.sh r4r ab s9 : --pipeline 1 --steps 10 --warmup 3 --reps 3
bash tools/gpu_run.sh r4r ab s9 : --workload prove_witness --steps 10 --warmup 3 --reps 3
bash tools/gpu_run.sh r4r ab s9 : --instance realistic --steps 10 --warmup 3 --reps 3
bash tools/gpu_run.sh r4r ab s9 : --workload msm_g1 --steps 40 --warmup 5 --reps 3
bash tools/gpu_run.sh r4r ab s9 : --workload msm_g1 --pipeline 1 --steps 40 --warmup 5 --reps 3
bash tools/gpu_run.sh r4r ab s9 : --log2n 22 --steps 4 --warmup 1 --reps 3
bash tools/gpu_run.sh r4r ab s9 : --log2n 19 --steps 20 --warmup 3 --reps 3
python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -3
