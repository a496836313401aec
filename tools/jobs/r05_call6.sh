# Round 5, sixth GPU call: the table-policy tests again (eviction assertion), then the mixed soak with round 5's operations: host-buffer
# tickets, in-place updates, and the table policy / release / build changing at random under whatever is in flight.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5f
mkdir -p gpurun_out/$T
(timeout 600 python -m pytest tests/test_gpu_table_policy.py tests/test_gpu_c_drivers.py -q 2>&1 | tail -15) > gpurun_out/$T/pytest_policy.txt; tail -3 gpurun_out/$T/pytest_policy.txt
(timeout 600 python tools/soak_mixed.py 200 11 2>&1 | tail -45) > gpurun_out/$T/soak_mixed.txt; tail -45 gpurun_out/$T/soak_mixed.txt
