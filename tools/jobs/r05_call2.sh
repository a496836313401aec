# Round 5, second GPU call: the eviction driver again, how a host ticket's arrays should be staged (A/B), real issue cycles per
# instruction class (tools/ubench_issue), and the table-free route at 2^22 and 2^24 constraints on one GPU.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5b
mkdir -p gpurun_out/$T
(timeout 300 python -m pytest tests/test_gpu_c_drivers.py -q -k "memory_eviction or stream_host" 2>&1 | tail -30) > gpurun_out/$T/pytest_cdrivers.txt; tail -3 gpurun_out/$T/pytest_cdrivers.txt
(timeout 120 ./tools/ubench_issue 2>&1) > gpurun_out/$T/ubench_issue.txt; cat gpurun_out/$T/ubench_issue.txt
(timeout 900 python tools/stream_host_ab.py 20 2>&1) > gpurun_out/$T/ab_host_stage.txt; cat gpurun_out/$T/ab_host_stage.txt
for pol in never always; do
  (timeout 600 python bench.py --log2n 22 --table-policy $pol --steps 4 --warmup 1 --reps 3 --no-extras --no-check --cpu-log2n 0 2>gpurun_out/$T/err_2p22_$pol.txt | tail -1) > gpurun_out/$T/bench_2p22_$pol.json
  python -c "import json;d=json.loads(open('gpurun_out/$T/bench_2p22_$pol.json').read());print('2^22 $pol', d['ms_per_step'], d['value'], d.get('memory'))"
done
(timeout 1500 python bench.py --log2n 24 --table-policy never --steps 3 --warmup 1 --reps 2 --settle-ms 0 --no-extras 2>gpurun_out/$T/err_2p24.txt | tail -1) > gpurun_out/$T/bench_2p24_never.json
tail -5 gpurun_out/$T/err_2p24.txt; head -c 1500 gpurun_out/$T/bench_2p24_never.json
