"""Per-kernel totals of a rocprofv3 counter_collection.csv.  With a second argument `mfma`: adds the MFMA-busy share
= SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs;
MI355X_MICROARCH.md, DVFS give-back), SIMDs = 256 CUs x 4."""
import csv, sys, collections
path = sys.argv[1]
mfma = len(sys.argv) > 2 and sys.argv[2] == "mfma"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"][:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:16]:
    print(k)
    for c, v in d.items():
        print("   %-28s total %.4g  per-dispatch %.4g  (n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
    if mfma and d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0
        print("   MFMA-busy share of the matrix pipes: %.1f %%" % (100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles)))
