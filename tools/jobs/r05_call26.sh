#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
T=r5copy4; mkdir -p gpurun_out/$T
for th in 8 1; do
GS_COPY_THREADS=$th GS_HOST_TRACE=1 GS_HOST_STAGE=1 timeout 600 python tools/stream_host_ab.py --child 20 2> gpurun_out/$T/host_trace_$th.txt | tail -1
grep "staged_h2d" gpurun_out/$T/host_trace_$th.txt | sed -n '100,112p'
done
