cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6x_repro
timeout 600 python tools/power_trace.py 4 20 --streams-only 2>&1 | grep "stream\|proofs\|idle\|alternating" | tee gpurun_out/r6x_repro/streams_with_temps.txt
