#!/bin/bash
# round 5, call 19: why does the second proof of a fresh key wait for the whole background build?
set -x
cd /root/repo; export TMPDIR=/tmp
T=r5bg2; mkdir -p gpurun_out/$T
{
echo "== default (slab 2^15, low-priority table stream)"; timeout 300 python tools/time_first_proof.py auto 20 8 | tail -2
echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 300 python tools/time_first_proof.py auto 20 8 | tail -2
echo "== GS_TABLE_STREAM_LOW=0"; GS_TABLE_STREAM_LOW=0 timeout 300 python tools/time_first_proof.py auto 20 8 | tail -2
echo "== GS_TABLE_STREAM_LOW=0 GPU_MAX_HW_QUEUES=8"; GS_TABLE_STREAM_LOW=0 GPU_MAX_HW_QUEUES=8 timeout 300 python tools/time_first_proof.py auto 20 8 | tail -2
echo "== slab 2^18 GPU_MAX_HW_QUEUES=8"; GS_TABLE_BG_SLAB_LOG2=18 GPU_MAX_HW_QUEUES=8 timeout 300 python tools/time_first_proof.py auto 20 8 | tail -2
} 2>&1 | grep -v "^+\|amdgpu.ids" | tee gpurun_out/$T/second_proof.txt
D=/tmp/prof_bg; rm -rf $D
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python /root/repo/tools/time_first_proof.py auto 20 4 > /root/repo/gpurun_out/$T/trace_run.txt 2>&1 )
TR=$(find $D -name "*kernel_trace.csv" | head -1)
python - "$TR" > gpurun_out/$T/trace_build_window.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
b = [r for r in rows if "k_build_table" in r["Kernel_Name"]]
t0, t1 = b[0]["s"], b[-1]["e"]
print("build window %.1f ms, %d build launches, queues %s" % ((t1 - t0) / 1e6, len(b), sorted(set(r["Queue_Id"] for r in b))))
n = 0
for r in rows:
    if r["e"] < t0 - 2_000_000 or r["s"] > t1 + 2_000_000: continue
    name = r["Kernel_Name"].split("(")[0].replace("void gs::", "").replace("gs::", "")[:44]
    if "k_build_table" in name and n > 60 and r is not b[-1]: continue
    n += 1
    print("%9.3f %8.3f  q%-2s %s" % ((r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e6, r["Queue_Id"], name))
PY
head -120 gpurun_out/$T/trace_build_window.txt
