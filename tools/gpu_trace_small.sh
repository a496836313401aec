#!/bin/bash
# kernel timelines of small blocking operations: one 2^16 G1 MSM and one 2^16 proof
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-small}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for wl in msm_g1 prove; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$wl -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --log2n 16 --steps 4 --warmup 2 --reps 1 --pipeline 1 --cpu-log2n 0 --no-check --no-extras > $OUT/bench_$wl.txt 2>&1
  F=$(find $OUT/prof_$wl -name "*kernel_trace.csv" | head -1)
  python - "$F" > $OUT/timeline_$wl.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
acc = [i for i, r in enumerate(rows) if "k_digits" in r["Kernel_Name"]]
# the last operation: from its first plan kernel on
start = acc[-1] if "msm" in sys.argv[1] or True else acc[-2]
import os
if "prove" in os.path.basename(os.path.dirname(os.path.dirname(sys.argv[1]))) or "prof_prove" in sys.argv[1]:
    start = acc[-2]
rows = rows[max(0, start - 2):]
t0 = rows[0]["s"]
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void gs::", "").replace("gs::", "")[:44]
    print("%9.3f %8.3f  q%-2s %-44s" % ((r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e6, r.get("Queue_Id", "?"), name))
PY
  rm -rf $OUT/prof_$wl
  tail -1 $OUT/bench_$wl.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$wl 2^16 blocking', round(d['ms_per_step'],3), 'ms', d['device_ms_per_step'])"
done
