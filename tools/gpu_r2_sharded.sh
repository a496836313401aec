#!/bin/bash
TAG=${1:-r02shard}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps(d, indent=1))"; }
( timeout 900 python bench.py --workload prove_sharded --logical-shards 8 --log2n 22 --steps 3 --warmup 1 --reps 3 --cpu-log2n 0 2>$OUT/sharded_stderr.txt | line ) > $OUT/bench_line_prove_sharded_8_logical_2p22.json
( timeout 900 python bench.py --workload msm_sharded --logical-shards 8 --log2n 22 --steps 5 --warmup 1 --reps 3 --cpu-log2n 0 2>>$OUT/sharded_stderr.txt | line ) > $OUT/bench_line_msm_sharded_8_logical_2p22.json
( timeout 900 python bench.py --workload prove_sharded --logical-shards 8 --log2n 20 --steps 5 --warmup 1 --reps 3 --cpu-log2n 0 2>>$OUT/sharded_stderr.txt | line ) > $OUT/bench_line_prove_sharded_8_logical_2p20.json
( timeout 900 python bench.py --log2n 22 --reps 3 --steps 5 --cpu-log2n 0 --no-extras 2>/dev/null | line ) > $OUT/bench_line_2p22_single.json
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/" + os.environ.get("TAGX", "")
PY
for f in $OUT/bench_line_*sharded*.json; do python -c "
import json,sys
d=json.load(open('$f')); s=d['sharding']; print('$f'.split('/')[-1], 'ms/step', round(d['ms_per_step'],2), 'per-shard', [round(x,2) for x in s['per_shard_ms']], 'gather', round(s['gather_ms_one_rank_rccl'],3), 'modelled', s['MODELLED_not_measured']['ms_per_step'], 'used_rccl', s['used_rccl'])"; done
