# Round 5, fourth GPU call: eviction driver, operand-diversity modes of ubench_issue, the real products in real cycles (ubench_mulmod).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5d
mkdir -p gpurun_out/$T
(timeout 300 python -m pytest tests/test_gpu_c_drivers.py -q -k memory_eviction 2>&1 | tail -30) > gpurun_out/$T/pytest_cdrivers.txt; tail -3 gpurun_out/$T/pytest_cdrivers.txt
(timeout 120 ./tools/ubench_issue 2>&1) > gpurun_out/$T/ubench_issue.txt; tail -8 gpurun_out/$T/ubench_issue.txt
(timeout 120 ./tools/ubench_mulmod 2>&1) > gpurun_out/$T/ubench_mulmod.txt; cat gpurun_out/$T/ubench_mulmod.txt
