"""Kernel timeline between the k-th and (k+m)-th G2 accumulation of a rocprofv3 kernel_trace.csv (dev tool)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
k, m = int(sys.argv[2]), int(sys.argv[3])
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
g2 = [r for r in rows if "k_bucket_accumulate<gs::Fq2Tag>" in r["Kernel_Name"]]
t0, t1 = g2[k]["s"], g2[k + m]["s"]
print("window: G2 accumulation #%d .. #%d of %d = %.3f ms (%.3f ms per proof)" % (k, k + m, len(g2), (t1 - t0) / 1e6, (t1 - t0) / 1e6 / m))
for r in rows:
    if r["e"] < t0 or r["s"] > t1:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void gs::", "").replace("gs::", "")[:44]
    d = (r["e"] - r["s"]) / 1e6
    if d < 0.03 and "accumulate" not in name:
        continue
    print("%9.3f %8.3f  q%-2s %-44s" % ((r["s"] - t0) / 1e6, d, r.get("Queue_Id", "?"), name))
