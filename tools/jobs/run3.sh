set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/gpurun_variants/lib_r3tails.so
bash tools/gpu_run.sh r4c tests
bash tools/gpu_run.sh r4c env GS_LIB=$OLD
for v in "" $OLD; do
  for r in 1 2; do
  echo -n "msm 2^16 blocking lib=${v:-new}: "; GS_LIB=$v python bench.py --workload msm_g1 --log2n 16 --pipeline 1 --steps 200 --warmup 20 --reps 3 --cpu-log2n 0 --no-extras --no-check 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_reps'], d.get('device_ms_per_step'))"
  echo -n "msm 2^20 blocking lib=${v:-new}: "; GS_LIB=$v python bench.py --workload msm_g1 --log2n 20 --pipeline 1 --steps 40 --warmup 5 --reps 3 --cpu-log2n 0 --no-extras --no-check 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_reps'], d.get('device_ms_per_step'))"
  done
done 2>&1 | tee gpurun_out/r4c/msm_blocking.txt
