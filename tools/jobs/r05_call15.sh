#!/bin/bash
# round 5, call 15: device-side chunk choice (k_choose_chunk) -- parity on the MSM/prove modules, then A/B GS_CHUNK_MODEL=0/1
set -x
cd /root/repo; export TMPDIR=/tmp
T=r5chunk; mkdir -p gpurun_out/$T
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_table_policy.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/$T/pytest.txt
cat gpurun_out/$T/pytest.txt
one() {  # name, env, args...
  local name=$1 envs=$2; shift 2
  env $envs timeout 600 python bench.py --cpu-log2n 0 --no-extras --no-check "$@" 2> gpurun_out/$T/err_$name.txt | tail -1 > gpurun_out/$T/bench_$name.json
  python - "$name" gpurun_out/$T/bench_$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read())
t=d["device_ms_per_step"]
print("%-28s %8.3f ms (reps %s) acc g1 %.2f g2 %.2f poly %.2f plan %.2f reduce %.2f heavy %s" % (sys.argv[1], d["ms_per_step"], ["%.2f"%x for x in d["ms_per_step_reps"]], t["acc_g1_ms"], t["acc_g2_ms"], t["poly_ms"], t["plan_ms"], t["reduce_ms"], d["plan_per_step"]["heavy_buckets"]))
PY
}
for m in 0 1; do
  one realistic_m$m GS_CHUNK_MODEL=$m --instance realistic --steps 10 --warmup 3 --reps 3
  one dense_m$m GS_CHUNK_MODEL=$m --steps 10 --warmup 2 --reps 3
  one gates_m$m GS_CHUNK_MODEL=$m --instance gates --steps 10 --warmup 2 --reps 3
  one real_wit_m$m GS_CHUNK_MODEL=$m --instance realistic --workload prove_witness --steps 10 --warmup 3 --reps 3
  one p2p18_m$m GS_CHUNK_MODEL=$m --log2n 18 --steps 40 --warmup 5 --reps 3
  one p2p22_m$m GS_CHUNK_MODEL=$m --log2n 22 --steps 4 --warmup 1 --reps 2
  one pin_m$m GS_CHUNK_MODEL=$m --workload prove_pinocchio --steps 10 --warmup 2 --reps 2
  one blocking_real_m$m GS_CHUNK_MODEL=$m --instance realistic --pipeline 1 --steps 10 --warmup 3 --reps 2
done 2>&1 | grep -v "^+" | tee gpurun_out/$T/ab_chunk_model.txt
