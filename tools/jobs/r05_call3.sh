# Round 5, third GPU call: the C drivers again, ubench_issue with the chain-count modes, the host-ticket streams with px staged up front.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5c
mkdir -p gpurun_out/$T
(timeout 300 python -m pytest tests/test_gpu_c_drivers.py tests/test_gpu_stream_host.py -q 2>&1 | tail -30) > gpurun_out/$T/pytest_cdrivers.txt; tail -3 gpurun_out/$T/pytest_cdrivers.txt
(timeout 120 ./tools/ubench_issue 2>&1) > gpurun_out/$T/ubench_issue.txt; cat gpurun_out/$T/ubench_issue.txt
(timeout 300 python tools/stream_host_ab.py --child 20 2>&1 | tail -1) > gpurun_out/$T/stream_mode1.txt; cat gpurun_out/$T/stream_mode1.txt
