# GS_MSM_TICKET_STREAMS: the by-size default (2) against 0 at 2^21 and 2^22 terms, two rounds, one box.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4u; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); t = d.get("device_ms_per_step", {})
        print("median %.3f min %.3f ms/step | value %.4g %s | acc g1 %.2f plan %.2f reduce %.2f" % (d["ms_per_step"], d.get("ms_per_step_min", 0),
              d["value"], d["unit"], t.get("acc_g1_ms", 0), t.get("plan_ms", 0), t.get("reduce_ms", 0)))
PY
}
qb() { python bench.py "$@" --cpu-log2n 0 --no-check --no-extras 2>&1 | tail -1 > $O/.last.json; line $O/.last.json; }
for round in 1 2; do
  for v in 0 2; do
    export GS_MSM_TICKET_STREAMS=$v
    echo -n "msm_g1 2^22 pipelined, ticket streams $v: "; qb --workload msm_g1 --log2n 22 --steps 12 --warmup 3 --reps 3
    echo -n "msm_g1 2^21 pipelined, ticket streams $v: "; qb --workload msm_g1 --log2n 21 --steps 20 --warmup 3 --reps 3
  done
done 2>&1 | tee $O/ab_msm_ticket_streams_large.txt
