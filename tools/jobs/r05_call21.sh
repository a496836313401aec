#!/bin/bash
# round 5, call 21: staging threads of the host-buffer tickets
cd /root/repo; export TMPDIR=/tmp
T=r5copy; mkdir -p gpurun_out/$T
for th in 4 8 16 2; do GS_AB_MODES=1 GS_COPY_THREADS=$th timeout 600 python tools/stream_host_ab.py 20; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$T/ab_copy_threads.txt
nproc; grep -c processor /proc/cpuinfo
