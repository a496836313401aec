#!/bin/bash
# round 5, call 18: background table builds in small slabs -- the warm-up transient of a fresh key under policy auto
set -x
cd /root/repo; export TMPDIR=/tmp
T=r5bg; mkdir -p gpurun_out/$T
for lg in 18 15 13; do
  echo "== GS_TABLE_BG_SLAB_LOG2=$lg"
  GS_TABLE_BG_SLAB_LOG2=$lg timeout 300 python tools/time_first_proof.py auto 20 40
  GS_TABLE_BG_SLAB_LOG2=$lg timeout 300 python tools/time_first_proof.py auto 20 40 | tail -1
  GS_TABLE_BG_SLAB_LOG2=$lg timeout 300 python tools/time_first_proof.py auto 16 40 | tail -1
done 2>&1 | grep -v "^+" | tee gpurun_out/$T/background_build_slabs.txt
timeout 300 python tools/time_first_proof.py always 20 6 2>&1 | tee -a gpurun_out/$T/background_build_slabs.txt
timeout 900 python -m pytest tests/test_gpu_table_policy.py -m gpu -q -x 2>&1 | tail -3
