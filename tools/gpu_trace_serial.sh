#!/bin/bash
# isolated duration of every kernel of one 2^20 proof: everything serialised on one stream (GS_NO_OVERLAP=1), blocking calls
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-serial}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
GS_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --reps 1 --pipeline 1 --cpu-log2n 0 --no-check --no-extras > $OUT/bench.txt 2>&1
F=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$F" > $OUT/timeline_serialised.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
g2 = [i for i, r in enumerate(rows) if "k_bucket_accumulate<gs::Fq2Tag>" in r["Kernel_Name"]]
# the last proof: from the k_digits before its G2 accumulation to the end
i0 = g2[-1]
while i0 > 0 and "k_digits" not in rows[i0]["Kernel_Name"]: i0 -= 1
rows = rows[max(0, i0 - 2):]
t0 = rows[0]["s"]
tot = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void gs::", "").replace("gs::", "")[:44]
    d = (r["e"] - r["s"]) / 1e6
    tot[name] = tot.get(name, 0) + d
    print("%9.3f %8.3f  %-44s" % ((r["s"] - t0) / 1e6, d, name))
print("--- per kernel, summed over the proof")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print("%8.3f  %s" % (v, k))
PY
rm -rf $OUT/prof
tail -25 $OUT/timeline_serialised.txt
