# Round 5, fifth GPU call: the whole GPU suite and the default bench line with the adopted host-ticket staging, rocprofv3 kernel stats
# of the bench command, a timeline of the realistic-witness workload, bench lines of the other sizes.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5e
mkdir -p gpurun_out/$T
(timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -15) > gpurun_out/$T/pytest_gpu.txt; tail -3 gpurun_out/$T/pytest_gpu.txt
(timeout 900 python bench.py 2>gpurun_out/$T/bench_err.txt | tail -1) > gpurun_out/$T/bench_line.json; head -c 400 gpurun_out/$T/bench_line.json
bash tools/gpu_run.sh $T stats bench_steps3 --steps 3 --warmup 1 --reps 1 --cpu-log2n 0 --no-extras --no-check
bash tools/gpu_run.sh $T trace realistic --instance realistic --steps 4 --warmup 2 --reps 1 --cpu-log2n 0 --no-extras --no-check
bash tools/gpu_run.sh $T bench realistic --instance realistic --steps 10 --warmup 3 --reps 3 --cpu-log2n 0 --no-extras -- bench 2p16 --log2n 16 --steps 100 --warmup 10 --cpu-log2n 0 --no-extras -- bench 2p18 --log2n 18 --steps 40 --warmup 5 --cpu-log2n 0 --no-extras -- bench witness_pipelined --workload prove_witness --steps 10 --warmup 2 --cpu-log2n 0 -- bench pinocchio --workload prove_pinocchio --steps 10 --warmup 2 --cpu-log2n 0 -- bench msm_g1 --workload msm_g1 --steps 40 --warmup 5 --cpu-log2n 0
