#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/profpx; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o px -- python $GRAFT_REPO_ROOT/tools/time_px.py 20 > $OUT/run.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/profpx/prof/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-70s calls %5s total %8.2f ms avg %8.3f ms" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
PY
rm -rf $OUT/prof
