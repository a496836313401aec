#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
T=r5copy5; mkdir -p gpurun_out/$T
for rep in 1 2 3; do
GS_HOST_STAGE=1 timeout 600 python tools/stream_host_ab.py --child 20 2>/dev/null | tail -1
done | tee gpurun_out/$T/one_sync.txt
timeout 900 python -m pytest tests/test_gpu_stream_host.py tests/test_gpu_c_drivers.py -m gpu -q -x 2>&1 | tail -3
