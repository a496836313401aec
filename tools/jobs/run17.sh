set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r4n env GS_PLANW_STREAM=3 GS_PLANW_STREAM=0 : --steps 10 --warmup 3 --reps 5
bash tools/gpu_run.sh r4n env GS_PLANW_STREAM=3 : --log2n 19 --steps 20 --warmup 3 --reps 3
python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "pipelined or witness or ticket" 2>&1 | tail -3
