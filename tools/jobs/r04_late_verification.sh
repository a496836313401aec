# Verification of the late round-4 state: the whole GPU suite, the default line, the 2^22-term MSM, the launcher-free 8-device line,
# the N-rank rehearsal.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r4final2
bash tools/gpu_run.sh $T tests
(python bench.py 2>&1 | tail -1) > gpurun_out/$T/bench_line.json; head -c 400 gpurun_out/$T/bench_line.json
bash tools/gpu_run.sh $T bench msm_g1_2p22 --workload msm_g1 --log2n 22 --steps 12 --warmup 3 --reps 3 --cpu-log2n 0 -- bench msm_g1 --workload msm_g1 --steps 40 --warmup 5 --cpu-log2n 0
(python bench.py --gpus 8 --steps 5 --warmup 2 --reps 2 2>&1 | tail -1) > gpurun_out/$T/bench_plain_8_logical.json; head -c 300 gpurun_out/$T/bench_plain_8_logical.json
bash tools/scale_selftest.sh 2 > gpurun_out/$T/selftest.txt 2>&1; tail -12 gpurun_out/$T/selftest.txt
