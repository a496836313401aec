"""Print the kernel timeline of the LAST bench step from a rocprofv3 kernel_trace.csv (dev tool)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# last step = from the last k_digits-after-k_copy_reversed group; simply take kernels after the last gap > 1.5 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i]["s"] - max(r["e"] for r in rows[max(0, i - 50):i]) > 1_000_000:
        cut = i
rows = rows[cut:]
t0 = rows[0]["s"]
print("kernels in last step:", len(rows), "span %.3f ms" % ((max(r["e"] for r in rows) - t0) / 1e6))
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:60]
    print("%9.3f %9.3f  q%-3s %-60s grid %s" % ((r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e6, r.get("Queue_Id", "?"), name, r.get("Grid_Size", "")))
