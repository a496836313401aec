#!/bin/bash
# round 5, call 16: sweep of the accumulate chunk size (development library: GS_CHUNK forces the host's chunk, GS_CHUNK_MODEL=0)
set -x
cd /root/repo; export TMPDIR=/tmp
T=r5chunk2; mkdir -p gpurun_out/$T
export GS_LIB=/root/repo/gpurun_variants/lib_dev.so GS_CHUNK_MODEL=0
one() {  # name, env, args...
  local name=$1 envs=$2; shift 2
  env $envs timeout 600 python bench.py --cpu-log2n 0 --no-extras --no-check "$@" 2> gpurun_out/$T/err_$name.txt | tail -1 > gpurun_out/$T/bench_$name.json
  python - "$name" gpurun_out/$T/bench_$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read())
t=d["device_ms_per_step"]
print("%-28s %8.3f ms (min %.3f) acc g1 %.2f g2 %.2f poly %.2f plan %.2f reduce %.2f heavy %s" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_min"], t["acc_g1_ms"], t["acc_g2_ms"], t["poly_ms"], t["plan_ms"], t["reduce_ms"], d["plan_per_step"]["heavy_buckets"]))
PY
}
for k in 8 12 16 20 24 28 32 36 40 48 64; do
  one dense_c$k GS_CHUNK=$k --steps 10 --warmup 2 --reps 3
  one realistic_c$k GS_CHUNK=$k --instance realistic --steps 10 --warmup 3 --reps 3
  one gates_c$k GS_CHUNK=$k --instance gates --steps 10 --warmup 2 --reps 3
  one msm_g1_c$k GS_CHUNK=$k --workload msm_g1 --steps 40 --warmup 5 --reps 3
  one p2p18_c$k GS_CHUNK=$k --log2n 18 --steps 40 --warmup 5 --reps 3
done 2>&1 | grep -v "^+" | tee gpurun_out/$T/sweep_chunk.txt
