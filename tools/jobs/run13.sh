set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r4k tests
bash tools/gpu_run.sh r4k env GS_PLANW_STREAM=0 : --workload prove_witness --steps 10 --warmup 3 --reps 3
bash tools/gpu_run.sh r4k env GS_PLANW_STREAM=0 : --log2n 22 --steps 4 --warmup 1 --reps 3
bash tools/gpu_run.sh r4k env GS_PLANW_STREAM=0 : --steps 10 --warmup 3 --reps 5
bash tools/gpu_run.sh r4k env GS_PLANW_STREAM=0 : --workload prove_witness --log2n 19 --steps 20 --warmup 3 --reps 3
