"""Per-kernel average of one rocprofv3 PMC counter (counter_collection.csv) -- dev tool."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
import os
min_grid = int(os.environ.get("PMC_MIN_GRID", "0"))      # only dispatches with at least this many work-items (e.g. the 2^21-point NTT passes)
for r in rows:
    if r.get("Counter_Name") != want:
        continue
    if min_grid and int(r.get("Grid_Size", "0") or 0) < min_grid:
        continue
    k = r["Kernel_Name"].split("(")[0]
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
print("kernel, dispatches, avg %s per dispatch (counter units: KB)" % want)
top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-60s %5d %14.1f" % (k[:60], n, v / n))
