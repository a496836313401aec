# Round 5, eleventh GPU call: HBM traffic of the table-free route (FETCH_SIZE / WRITE_SIZE through bench.py's own live PMC passes):
# every window gathers from the SAME 64 MiB base array instead of 15 different table rows.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=r5k
mkdir -p gpurun_out/$T
(timeout 600 python bench.py --table-policy never --cpu-log2n 0 --no-check --steps 10 --warmup 3 --reps 3 2>gpurun_out/$T/err.txt | tail -1) > gpurun_out/$T/bench_table_free.json
python -c "
import json;d=json.loads(open('gpurun_out/$T/bench_table_free.json').read())
print(d['ms_per_step'], d['roofline'], d.get('memory'))"
