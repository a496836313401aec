#!/bin/bash
# round 5, call 20: background builds after the scratch fix -- slab sizes, the whole transient
set -x
cd /root/repo; export TMPDIR=/tmp
T=r5bg3; mkdir -p gpurun_out/$T
{
for lg in 18 17 16 15 14; do
  echo "== GS_TABLE_BG_SLAB_LOG2=$lg"
  GS_TABLE_BG_SLAB_LOG2=$lg timeout 300 python tools/time_first_proof.py auto 20 40
done
echo "== 2^16 key"; timeout 300 python tools/time_first_proof.py auto 16 30 | tail -2
echo "== always"; timeout 300 python tools/time_first_proof.py always 20 6
} 2>&1 | grep -v "^+\|amdgpu.ids" | tee gpurun_out/$T/background_build_slabs.txt
timeout 900 python -m pytest tests/test_gpu_table_policy.py tests/test_gpu_stream_host.py -m gpu -q -x 2>&1 | tail -3
