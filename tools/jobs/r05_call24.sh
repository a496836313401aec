#!/bin/bash
# round 5, call 24: staging piece size / buffers / copy threads of the host-buffer tickets (host times from GS_HOST_TRACE)
cd /root/repo; export TMPDIR=/tmp
T=r5copy2; mkdir -p gpurun_out/$T
run() {
  env "$@" GS_HOST_TRACE=1 GS_HOST_STAGE=1 timeout 600 python tools/stream_host_ab.py --child 20 2> gpurun_out/$T/host_trace.txt | tail -1 > gpurun_out/$T/line.json
  python - "$*" <<'PY'
import re, sys, json
L=[l for l in open("/root/repo/gpurun_out/r5copy2/host_trace.txt") if "begin:" in l]
beg=[tuple(map(float,re.findall(r"stage w ([\d.]+) ms, stage px ([\d.]+) ms, enqueue ([\d.]+)", l)[0])) for l in L][16:64]
f=lambda xs: sum(xs)/max(len(xs),1)
d=json.loads(open("/root/repo/gpurun_out/r5copy2/line.json").read())
print("%-58s stage w %.2f px %.2f enqueue %.2f ms | witness_host %.2f px_host %.2f resident %.2f px_resident %.2f" % (sys.argv[1], f([x[0] for x in beg]), f([x[1] for x in beg]), f([x[2] for x in beg]), d["witness_host"], d["px_host"], d["resident"], d["px_resident_same_witness"]))
PY
}
{
run GS_STAGE_MIB=4 GS_STAGE_BUFFERS=2 GS_COPY_THREADS=4
run GS_STAGE_MIB=16 GS_STAGE_BUFFERS=3 GS_COPY_THREADS=8
run GS_STAGE_MIB=8 GS_STAGE_BUFFERS=3 GS_COPY_THREADS=8
run GS_STAGE_MIB=32 GS_STAGE_BUFFERS=3 GS_COPY_THREADS=8
run GS_STAGE_MIB=16 GS_STAGE_BUFFERS=2 GS_COPY_THREADS=8
run GS_STAGE_MIB=16 GS_STAGE_BUFFERS=3 GS_COPY_THREADS=4
run GS_STAGE_MIB=16 GS_STAGE_BUFFERS=3 GS_COPY_THREADS=16
run GS_STAGE_MIB=16 GS_STAGE_BUFFERS=3 GS_COPY_THREADS=1
run GS_STAGE_MIB=4 GS_STAGE_BUFFERS=2 GS_COPY_THREADS=4
run GS_STAGE_MIB=16 GS_STAGE_BUFFERS=3 GS_COPY_THREADS=8
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$T/ab_stage_pieces.txt
