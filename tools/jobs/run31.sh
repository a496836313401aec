# Depth of the block reduce at small sizes: GS_REDUCE_L = 1 / 2 / 4 (dev build), blocking and pipelined, two rounds, one box.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4v; mkdir -p $O
export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_dev.so
line() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); t = d.get("device_ms_per_step", {})
        print("median %.3f min %.3f ms/step | value %.4g %s | acc g1 %.2f g2 %.2f plan %.2f reduce %.2f" % (d["ms_per_step"], d.get("ms_per_step_min", 0),
              d["value"], d["unit"], t.get("acc_g1_ms", 0), t.get("acc_g2_ms", 0), t.get("plan_ms", 0), t.get("reduce_ms", 0)))
PY
}
qb() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 > $O/.last.json; line $O/.last.json; }
for round in 1 2; do
  for v in 4 2 1; do
    export GS_REDUCE_L=$v
    echo -n "msm_g1 2^16 blocking,  L $v: "; qb --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3 --pipeline 1
    echo -n "msm_g1 2^16 pipelined, L $v: "; qb --workload msm_g1 --log2n 16 --steps 200 --warmup 20 --reps 3
    echo -n "msm_g1 2^18 blocking,  L $v: "; qb --workload msm_g1 --log2n 18 --steps 100 --warmup 10 --reps 3 --pipeline 1
    echo -n "prove 2^16 blocking,   L $v: "; qb --log2n 16 --steps 100 --warmup 10 --reps 3 --pipeline 1
    echo -n "prove 2^16 pipelined,  L $v: "; qb --log2n 16 --steps 100 --warmup 10 --reps 3
    if [ $round = 1 ]; then echo -n "prove 2^20 blocking,   L $v: "; qb --steps 8 --warmup 2 --reps 3 --pipeline 1; fi
  done
done 2>&1 | tee $O/ab_reduce_depth.txt
