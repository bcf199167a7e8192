"""One train step as a kernel timeline (start offset, duration, gap to the previous kernel): the rocprofv3 kernel-trace CSV of
`bench.py --only-train-steps` (bash profiles/scripts/timeline.sh <precision> <batch>)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "mask_prepare" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
tot_gap = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("adn::", "").split("(")[0][:60]
    gap = (s - prev) / 1e3; tot_gap += max(gap, 0)
    print("%8.1f us  dur %7.1f  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev = e
print("step span %.1f us, %d launches, gaps %.1f us" % ((prev - t0) / 1e3, b - a, tot_gap))
