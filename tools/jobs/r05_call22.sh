#!/bin/bash
# round 5, call 22: is the px_host stream host-bound?  begin / collect host times (GS_HOST_TRACE) while streaming distinct witnesses
cd /root/repo; export TMPDIR=/tmp
T=r5copy; mkdir -p gpurun_out/$T
GS_HOST_TRACE=1 GS_HOST_STAGE=1 timeout 600 python tools/stream_host_ab.py --child 20 2> gpurun_out/$T/host_trace.txt | tail -1
python - <<'PY'
import re
L=[l for l in open("/root/repo/gpurun_out/r5copy/host_trace.txt") if "[gs host]" in l]
beg=[tuple(map(float,re.findall(r"stage w ([\d.]+) ms, stage px ([\d.]+) ms, enqueue ([\d.]+)", l)[0])) for l in L if "begin:" in l]
col=[tuple(map(float,re.findall(r"wait ([\d.]+) ms, fold ([\d.]+)", l)[0])) for l in L if "collect:" in l]
print(len(beg), "begins", len(col), "collects")
for i in range(0, len(beg), 16):
    b = beg[i:i+16]; c = col[i:i+16]
    f = lambda xs: sum(xs)/max(len(xs),1)
    print("ops %4d..: stage w %.3f  stage px %.3f  enqueue %.3f | collect wait %.3f fold %.3f" % (i, f([x[0] for x in b]), f([x[1] for x in b]), f([x[2] for x in b]), f([x[0] for x in c]), f([x[1] for x in c])))
PY
