# Wave issue priority (s_setprio) of the plan / polynomial / tail kernels: in-tree library (all 0) vs variants, two rounds, one box.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); t = d.get("device_ms_per_step", {})
        print("median %.3f min %.3f ms/step | value %.4g %s | acc g1 %.2f g2 %.2f poly %.2f plan %.2f reduce %.2f" % (d["ms_per_step"], d.get("ms_per_step_min", 0),
              d["value"], d["unit"], t.get("acc_g1_ms", 0), t.get("acc_g2_ms", 0), t.get("poly_ms", 0), t.get("plan_ms", 0), t.get("reduce_ms", 0)))
PY
}
qb() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 > $O/.last.json; line $O/.last.json; }
for round in 1 2; do
  for v in base plan3 plan3poly3 plan3poly3tail1 plan2poly1; do
    if [ $v = base ]; then unset GS_LIB; else export GS_LIB=$GRAFT_REPO_ROOT/gpurun_variants/lib_$v.so; fi
    echo -n "prove 2^20 pipelined, $v: "; qb --steps 10 --warmup 3 --reps 5
    echo -n "msm_g1 2^20 pipelined, $v: "; qb --workload msm_g1 --steps 40 --warmup 5 --reps 3
    echo -n "prove 2^16 pipelined, $v: "; qb --log2n 16 --steps 100 --warmup 10 --reps 3
    if [ $round = 1 ]; then echo -n "prove 2^20 blocking, $v: "; qb --steps 8 --warmup 2 --reps 3 --pipeline 1; fi
  done
done 2>&1 | tee $O/ab_wave_priority.txt
