#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
T=r5copy5; mkdir -p gpurun_out/$T
for cfg in "GS_STAGE_MIB=4 GS_STAGE_BUFFERS=2" "GS_STAGE_MIB=16" "GS_STAGE_MIB=8" "GS_STAGE_MIB=4 GS_STAGE_BUFFERS=3" "GS_STAGE_MIB=16 GS_COPY_THREADS=1"; do
echo -n "$cfg: "; env $cfg GS_HOST_STAGE=1 timeout 600 python tools/stream_host_ab.py --child 20 2>/dev/null | tail -1
done | tee gpurun_out/$T/update_regression.txt
