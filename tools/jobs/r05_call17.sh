#!/bin/bash
# round 5, call 17: shrink-only device chunk choice A/B (two rounds), steady-state timeline of the realistic instance with queue ids
set -x
cd /root/repo; export TMPDIR=/tmp
T=r5chunk3; mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_msm.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/$T/pytest.txt
cat gpurun_out/$T/pytest.txt
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
for round in 1 2; do for m in 0 1; do
  one realistic_m${m}_r$round GS_CHUNK_MODEL=$m --instance realistic --steps 10 --warmup 3 --reps 3
  one dense_m${m}_r$round GS_CHUNK_MODEL=$m --steps 10 --warmup 2 --reps 3
  one gates_m${m}_r$round GS_CHUNK_MODEL=$m --instance gates --steps 10 --warmup 2 --reps 3
  one real_wit_m${m}_r$round GS_CHUNK_MODEL=$m --instance realistic --workload prove_witness --steps 10 --warmup 3 --reps 3
  one p2p18_m${m}_r$round GS_CHUNK_MODEL=$m --log2n 18 --steps 40 --warmup 5 --reps 3
  one pin_m${m}_r$round GS_CHUNK_MODEL=$m --workload prove_pinocchio --steps 10 --warmup 2 --reps 2
  one blocking_real_m${m}_r$round GS_CHUNK_MODEL=$m --instance realistic --pipeline 1 --steps 10 --warmup 3 --reps 2
  one gates_p18_m${m}_r$round GS_CHUNK_MODEL=$m --instance gates --log2n 18 --steps 40 --warmup 5 --reps 3
done; done 2>&1 | grep -v "^+" | tee gpurun_out/$T/ab_chunk_model.txt
D=/tmp/prof_real; rm -rf $D
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python /root/repo/bench.py --instance realistic --steps 12 --warmup 3 --reps 1 --cpu-log2n 0 --no-extras --no-check > /root/repo/gpurun_out/$T/trace_run.txt 2>&1 )
TR=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/trace_steady.py "$TR" "k_bucket_accumulate<gs::Fq2Tag>" 8 2 > gpurun_out/$T/steady_realistic.txt
tail -30 gpurun_out/$T/steady_realistic.txt
