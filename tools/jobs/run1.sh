set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python bench.py --gpus 2 --steps 3 --warmup 1 --reps 1 --log2n 16) > gpurun_out/r4_plain2_2p16.log 2>&1; tail -c 3000 gpurun_out/r4_plain2_2p16.log
(time GS_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --reps 1 --log2n 16 --cpu-log2n 0) > gpurun_out/r4_ranks2_share_2p16.log 2>&1; tail -c 3000 gpurun_out/r4_ranks2_share_2p16.log
(time python bench.py --gpus 8 --steps 5 --warmup 2 --reps 2) > gpurun_out/r4_plain8_2p20.log 2>&1; tail -c 4000 gpurun_out/r4_plain8_2p20.log
(time python bench.py) > gpurun_out/r4_default.log 2>&1; tail -c 6000 gpurun_out/r4_default.log
