set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r4final2
mkdir -p gpurun_out/$T
(python bench.py 2>&1 | tail -1) > gpurun_out/$T/bench_line.json; head -c 300 gpurun_out/$T/bench_line.json
(python bench.py --instance realistic --cpu-log2n 0 --reps 3 2>&1 | tail -1) > gpurun_out/$T/bench_realistic.json; head -c 300 gpurun_out/$T/bench_realistic.json
bash tools/gpu_run.sh $T bench witness_pipelined --workload prove_witness --steps 10 --warmup 2 --cpu-log2n 0 -- bench witness_blocking --workload prove_witness --pipeline 1 --steps 10 --warmup 2 --cpu-log2n 0 -- bench 2p22 --log2n 22 --steps 4 --warmup 1 --reps 3 --cpu-log2n 0 --no-extras -- bench 2p18 --log2n 18 --steps 40 --warmup 5 --cpu-log2n 0 --no-extras -- bench 2p16 --log2n 16 --steps 100 --warmup 10 --cpu-log2n 0 --no-extras
bash tools/gpu_run.sh $T stats bench_steps3 --steps 3 --warmup 1 --reps 1 --cpu-log2n 0 --no-extras --no-check
