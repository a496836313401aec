"""Steady-state kernel timeline from a rocprofv3 kernel_trace.csv: the window between the k-th and (k+m)-th launch of the kernel whose name
contains `needle` (dev tool; tools/gpu_run.sh's `trace` action shows the LAST operation, which runs on a draining pipeline).
    python tools/trace_steady.py <kernel_trace.csv> <needle> <k> <m>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
needle, k, m = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
hits = [r for r in rows if needle in r["Kernel_Name"]]
t0, t1 = hits[k]["s"], hits[k + m]["s"]
print("window: launch #%d .. #%d of %d x '%s' = %.3f ms (%.3f ms per launch)" % (k, k + m, len(hits), needle, (t1 - t0) / 1e6, (t1 - t0) / 1e6 / m))
tot = {}
for r in rows:
    if r["e"] < t0 or r["s"] > t1:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void gs::", "").replace("gs::", "")[:48]
    d = (r["e"] - r["s"]) / 1e6
    tot[name] = tot.get(name, 0) + d
    print("%9.3f %8.3f  q%-2s %s" % ((r["s"] - t0) / 1e6, d, r.get("Queue_Id", "?"), name))
print("--- per kernel, summed over the window")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("%8.3f  %s" % (v, n))
