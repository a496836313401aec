set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r4final
bash tools/gpu_run.sh $T tests
(python bench.py 2>&1 | tail -1) > gpurun_out/$T/bench_line.json; head -c 400 gpurun_out/$T/bench_line.json
bash tools/gpu_run.sh $T bench 2p16 --log2n 16 --steps 100 --warmup 10 --cpu-log2n 0 --no-extras -- bench 2p18 --log2n 18 --steps 40 --warmup 5 --cpu-log2n 0 --no-extras -- bench 2p22 --log2n 22 --steps 4 --warmup 1 --reps 3 --cpu-log2n 0 --no-extras
bash tools/gpu_run.sh $T bench msm_g1 --workload msm_g1 --steps 40 --warmup 5 --cpu-log2n 0 -- bench msm_g1_2p16 --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --cpu-log2n 0 -- bench msm_g1_2p16_blocking --workload msm_g1 --log2n 16 --pipeline 1 --steps 200 --warmup 20 --cpu-log2n 0
bash tools/gpu_run.sh $T bench pinocchio --workload prove_pinocchio --steps 10 --warmup 2 --cpu-log2n 0 -- bench witness_pipelined --workload prove_witness --steps 10 --warmup 2 --cpu-log2n 0 -- bench witness_blocking --workload prove_witness --pipeline 1 --steps 10 --warmup 2 --cpu-log2n 0 -- bench blocking --pipeline 1 --steps 10 --warmup 2 --cpu-log2n 0 --no-extras
bash tools/gpu_run.sh $T stats bench_steps3 --steps 3 --warmup 1 --reps 1 --cpu-log2n 0 --no-extras --no-check
bash tools/gpu_run.sh $T trace pipelined --steps 4 --warmup 2 --reps 1 --cpu-log2n 0 --no-extras --no-check -- trace blocking --pipeline 1 --steps 3 --warmup 1 --reps 1 --cpu-log2n 0 --no-extras --no-check -- trace msm_2p16_blocking --workload msm_g1 --log2n 16 --pipeline 1 --steps 5 --warmup 2 --reps 1 --cpu-log2n 0 --no-extras --no-check -- trace prove_2p16 --log2n 16 --steps 6 --warmup 2 --reps 1 --cpu-log2n 0 --no-extras --no-check
bash tools/gpu_run.sh $T pmc sq k_bucket_accumulate --steps 1 --warmup 0 --reps 1 --settle-ms 0 --cpu-log2n 0 --no-extras --no-check : SQ_INSTS_VALU SQ_WAVES : SQ_BUSY_CYCLES GRBM_GUI_ACTIVE : SQ_INSTS_SALU SQ_INSTS_VMEM_RD
(python bench.py --gpus 8 --steps 5 --warmup 2 --reps 2 2>&1 | tail -1) > gpurun_out/$T/bench_plain_8_logical.json; head -c 300 gpurun_out/$T/bench_plain_8_logical.json
bash tools/scale_selftest.sh 2 > gpurun_out/$T/selftest.txt 2>&1; tail -12 gpurun_out/$T/selftest.txt
