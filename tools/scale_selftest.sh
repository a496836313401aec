#!/bin/bash
# What the first real multi-GPU run will execute, exercised end to end before anybody has the hardware:
#   bash tools/scale_selftest.sh [N]          (default N = 2; run it on the GPU box through gpurun)
# 1. the driver's own command line at N ranks -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- with
#    every rank mapped to GPU 0 (GS_BENCH_SHARE_GPU=1: gloo for the barriers and the max-over-ranks reduction; the library calls,
#    the weak-scaling bookkeeping and the ONE JSON line are exactly the multi-GPU path);
# 2. the strong-scaling workloads in rank mode on one rank (RCCL communicator from gs_comm_unique_id / gs_comm_init_rank,
#    ncclAllGather of the 416-byte records, gs_scalars_scatter's ncclSend/ncclRecv group): tests/test_gpu_zy_multi.py;
# 3. with more than one visible GPU: the same command line WITHOUT the sharing switch (RCCL over N physical devices:
#    ncclCommInitRank with nranks > 1 and the in-library gather between processes) for prove and prove_sharded.
# 0. (round 4) the PLAIN form the driver uses at N = 1 -- `python bench.py --gpus N`, no launcher, WORLD_SIZE unset -- which drives the
#    N devices from one process (logical devices on the visible GPUs when there are fewer than N) and must print ONE line with
#    `value` (weak), `strong.prove_sharded_*` (both routes), `strong.msm_sharded_2^22`, `rccl` and `devices`.
# Prints one line per step; exit status != 0 when a step fails.
set -u
N=${1:-2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
fail=0
step() {   # step "<title>" <share:0|1> <bench args...>: the driver's command line, one JSON line expected
  local title=$1 share=$2; shift 2
  echo "== $title"
  local out
  if ! out=$(GS_BENCH_SHARE_GPU=$share python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
             --master-port $((29500 + RANDOM % 200)) bench.py --gpus "$N" "$@" 2>&1); then
    echo "   FAILED (exit status):"; echo "$out" | tail -5; fail=1; return
  fi
  echo "$out" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   n_gpus', d['n_gpus'], '|', d['scaling'], '| %.4g %s' % (d['value'], d['unit']), '| %.3f ms/step' % d['ms_per_step'], '|', d['config']['workload'])" \
    || { echo "   FAILED (no JSON line):"; echo "$out" | tail -5; fail=1; }
}

plain() {  # plain "<title>" <bench args...>
  local title=$1; shift
  echo "== $title"
  local out
  if ! out=$(env -u WORLD_SIZE -u RANK -u LOCAL_RANK python bench.py --gpus "$N" "$@" 2>&1); then
    echo "   FAILED (exit status):"; echo "$out" | tail -5; fail=1; return
  fi
  echo "$out" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
s=d['strong']; k=[x for x in s if x.startswith('prove_sharded')][0]
need=[s[k]['px_route']['proof_equals_single_device'], s[k]['values_route']['proof_equals_single_device'], s['msm_sharded_2^22']['equals_naive_loop_golden']]
print('   n_gpus', d['n_gpus'], '| weak %.4g %s' % (d['value'], d['unit']), '| strong px %.2f ms, values %.2f ms, msm 2^22 %.2f ms' % (s[k]['px_route']['ms_per_step'], s[k]['values_route']['ms_per_step'], s['msm_sharded_2^22']['ms_per_step']), '| rccl', d['rccl'].get('ranks_seen'), 'rank(s),', d['rccl'].get('collectives'), 'collectives | checks', need)
assert d['n_gpus']==$N and all(need) and 'devices' in d and 'rccl' in d
" || { echo "   FAILED (line incomplete):"; echo "$out" | tail -5; fail=1; }
}
plain "plain form, one process over $N devices: weak + strong + rccl in ONE line" --steps 3 --warmup 1 --reps 1 --log2n 16
step "driver command line, $N ranks sharing GPU 0 (gloo): weak scaling, independent proofs" 1 --steps 3 --warmup 1 --reps 1 --log2n 16 --cpu-log2n 0 --no-extras
step "same, ONE proof sharded over the $N ranks (host gather): strong scaling" 1 --workload prove_sharded --steps 3 --warmup 1 --reps 1 --log2n 16 --cpu-log2n 0
echo "== rank-mode entry points at world size 1 (RCCL communicator, gather, scatter)"
python -m pytest tests/test_gpu_zy_multi.py -m gpu -x -q -k "world_1 or rccl or values_route" 2>&1 | tail -2 || fail=1
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
if [ "$NGPU" -ge "$N" ] && [ "$N" -gt 1 ]; then
  step "driver command line on $N physical GPUs (RCCL): weak scaling" 0 --steps 5 --warmup 2 --reps 1 --cpu-log2n 0 --no-extras
  step "ONE proof sharded over $N physical GPUs, in-library RCCL gather" 0 --workload prove_sharded --steps 5 --warmup 2 --reps 1 --cpu-log2n 0
  step "same, values route (owner polynomial stage, ncclSend/ncclRecv scatter)" 0 --workload prove_sharded --sharded-route values --steps 6 --warmup 2 --reps 1 --cpu-log2n 0
else
  echo "== $NGPU GPU(s) visible: the RCCL-between-processes steps need $N (skipped, NOT tested here)"
fi
exit $fail
