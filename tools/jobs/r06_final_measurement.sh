# Round 6, final measurement call: the whole GPU suite, smoke(), the default bench line (with the live PMC passes), rocprofv3 kernel stats of the
# bench command, the other instances and sizes, the 8-logical-device rehearsal, 2^24 and the attempt at 2^25.
#     gpurun --timeout 3000 -- 'bash tools/jobs/r06_final_measurement.sh [part]'     part: all (default) | suite | lines | large | scale
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r6final
PART=${1:-all}
mkdir -p gpurun_out/$T
if [ $PART = all ] || [ $PART = suite ]; then
  (timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -8) > gpurun_out/$T/pytest_gpu.txt; tail -3 gpurun_out/$T/pytest_gpu.txt
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/$T/smoke.txt; cat gpurun_out/$T/smoke.txt
fi
if [ $PART = all ] || [ $PART = lines ]; then
  (timeout 900 python bench.py 2>gpurun_out/$T/bench_err.txt | tail -1) > gpurun_out/$T/bench_line.json; head -c 300 gpurun_out/$T/bench_line.json; echo
  bash tools/gpu_run.sh $T stats bench_steps3 --steps 3 --warmup 1 --reps 1 --cpu-log2n 0 --no-extras --no-check
  (timeout 600 python bench.py --instance gates --cpu-log2n 13 --no-extras 2>/dev/null | tail -1) > gpurun_out/$T/bench_line_gates.json; head -c 300 gpurun_out/$T/bench_line_gates.json; echo
  bash tools/gpu_run.sh $T bench 2p16 --log2n 16 --steps 100 --warmup 10 --cpu-log2n 0 --no-extras -- bench 2p18 --log2n 18 --steps 40 --warmup 5 --cpu-log2n 0 --no-extras -- bench 2p22 --log2n 22 --steps 4 --warmup 1 --reps 3 --cpu-log2n 0 --no-extras -- bench realistic --instance realistic --steps 10 --warmup 3 --reps 3 --cpu-log2n 0 --no-extras -- bench witness_pipelined --workload prove_witness --steps 10 --warmup 2 --cpu-log2n 0 -- bench pinocchio --workload prove_pinocchio --steps 10 --warmup 2 --cpu-log2n 0 -- bench msm_g1 --workload msm_g1 --steps 40 --warmup 5 --cpu-log2n 0 -- bench msm_g1_2p16_blocking --workload msm_g1 --log2n 16 --pipeline 1 --steps 200 --warmup 20 --cpu-log2n 0
fi
if [ $PART = all ] || [ $PART = scale ]; then
  (timeout 300 python bench.py --gpus 8 --steps 5 --warmup 2 --reps 2 2>/dev/null | tail -1) > gpurun_out/$T/bench_plain_8_logical.json; head -c 300 gpurun_out/$T/bench_plain_8_logical.json; echo
  (timeout 1200 bash tools/scale_selftest.sh 8 2>&1 | tail -40) | tee gpurun_out/$T/scale_selftest_8.txt
  timeout 600 python tools/soak_mixed.py 150 7 2>&1 | tail -6 | tee gpurun_out/$T/soak_mixed.txt
fi
if [ $PART = check25 ]; then   # 2^25 once more WITH the closed-form checker leg (cpu-log2n > 0 enables it; the CPU baseline itself runs at 2^10)
  (timeout 3000 python bench.py --log2n 25 --table-policy never --steps 2 --warmup 1 --reps 1 --cpu-log2n 10 --no-extras 2>gpurun_out/$T/bench_2p25_checked_err.txt | tail -1) > gpurun_out/$T/bench_2p25_checked.json; head -c 300 gpurun_out/$T/bench_2p25_checked.json; echo; tail -3 gpurun_out/$T/bench_2p25_checked_err.txt
fi
if [ $PART = all ] || [ $PART = large ]; then
  (timeout 900 python bench.py --log2n 24 --table-policy never --steps 3 --warmup 1 --reps 2 --cpu-log2n 0 --no-extras 2>gpurun_out/$T/bench_2p24_err.txt | tail -1) > gpurun_out/$T/bench_2p24.json; head -c 400 gpurun_out/$T/bench_2p24.json; echo
  (timeout 1500 python bench.py --log2n 25 --table-policy never --steps 2 --warmup 1 --reps 1 --cpu-log2n 0 --no-extras 2>gpurun_out/$T/bench_2p25_err.txt | tail -1) > gpurun_out/$T/bench_2p25.json; head -c 400 gpurun_out/$T/bench_2p25.json; echo; tail -5 gpurun_out/$T/bench_2p25_err.txt
fi
