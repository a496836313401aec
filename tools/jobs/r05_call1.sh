# Round 5, first GPU call: the new entry points (host-buffer tickets, table policy, eviction), the whole GPU suite on both table
# routes, the default bench line with the new extras, and the SQ issue breakdown of the G1 accumulation.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5a
mkdir -p gpurun_out/$T
(timeout 900 python -m pytest tests/test_gpu_stream_host.py tests/test_gpu_table_policy.py tests/test_gpu_c_drivers.py -q --maxfail=12 2>&1 | tail -60) > gpurun_out/$T/pytest_new.txt
tail -5 gpurun_out/$T/pytest_new.txt
(timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --deselect tests/test_gpu_stream_host.py --deselect tests/test_gpu_table_policy.py --deselect tests/test_gpu_c_drivers.py 2>&1 | tail -60) > gpurun_out/$T/pytest_rest.txt
tail -5 gpurun_out/$T/pytest_rest.txt
(timeout 900 python bench.py 2>gpurun_out/$T/bench_err.txt | tail -1) > gpurun_out/$T/bench_line.json
head -c 600 gpurun_out/$T/bench_line.json; tail -5 gpurun_out/$T/bench_err.txt
bash tools/gpu_run.sh $T pmc sq_issue k_bucket_accumulate --steps 1 --warmup 0 --reps 1 --settle-ms 0 --cpu-log2n 0 --no-extras --no-check : SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU : SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA : SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD : SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM
