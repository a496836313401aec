#!/bin/bash
# round 5, call 25: copy workers that watch for the next job (no futex wake-up inside a staged upload); smaller proofs from the host
cd /root/repo; export TMPDIR=/tmp
T=r5copy3; mkdir -p gpurun_out/$T
run() {
  local logn=$1; shift
  env "$@" GS_HOST_TRACE=1 GS_HOST_STAGE=1 timeout 600 python tools/stream_host_ab.py --child $logn 2> gpurun_out/$T/host_trace.txt | tail -1 > gpurun_out/$T/line.json
  python - "2^$logn $*" <<'PY'
import re, sys, json
L=[l for l in open("/root/repo/gpurun_out/r5copy3/host_trace.txt") if "begin:" in l]
beg=[tuple(map(float,re.findall(r"stage w ([\d.]+) ms, stage px ([\d.]+) ms, enqueue ([\d.]+)", l)[0])) for l in L][16:64]
f=lambda xs: sum(xs)/max(len(xs),1)
d=json.loads(open("/root/repo/gpurun_out/r5copy3/line.json").read())
print("%-50s stage w %.2f px %.2f enqueue %.2f ms | witness_host %.2f px_host %.2f resident %.2f px_resident %.2f" % (sys.argv[1], f([x[0] for x in beg]), f([x[1] for x in beg]), f([x[2] for x in beg]), d["witness_host"], d["px_host"], d["resident"], d["px_resident_same_witness"]))
PY
}
{
run 20 GS_COPY_THREADS=8
run 20 GS_COPY_THREADS=16
run 20 GS_COPY_THREADS=4
run 20 GS_COPY_THREADS=8 GS_STAGE_MIB=8
run 20 GS_COPY_THREADS=8
run 18 GS_COPY_THREADS=8
run 18 GS_COPY_THREADS=1
run 16 GS_COPY_THREADS=8
run 16 GS_COPY_THREADS=1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$T/ab_copy_spin.txt
timeout 900 python -m pytest tests/test_gpu_stream_host.py -m gpu -q -x 2>&1 | tail -3
