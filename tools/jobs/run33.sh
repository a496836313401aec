# Mixed soak of the final round-4 library: random interleavings of every pipelined operation, each result against its blocking twin.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4w
timeout 560 python tools/soak_mixed.py 400 1 > gpurun_out/r4w/soak_mixed.txt 2>&1; echo "exit $?" >> gpurun_out/r4w/soak_mixed.txt; tail -30 gpurun_out/r4w/soak_mixed.txt
