# Round 5, seventh GPU call: why do host-buffer tickets with w + px cost 6-9 % (and the event-ordered copy 12 %)?  The copy stream is the
# fifth stream of a context (beyond four, two share a hardware queue) and H2D copies may run as blit kernels: A/B of the runtime's
# switches on the distinct-witness streams.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5g
mkdir -p gpurun_out/$T
for v in "" "GPU_MAX_HW_QUEUES=8" "HSA_ENABLE_SDMA=0" "GS_HOST_STAGE=0" "GS_HOST_STAGE=0 GPU_MAX_HW_QUEUES=8"; do
  echo "== env: ${v:-default}" | tee -a gpurun_out/$T/ab_runtime_switches.txt
  (env $v timeout 300 python tools/stream_host_ab.py --child 20 2>&1 | tail -1) | tee -a gpurun_out/$T/ab_runtime_switches.txt
done
