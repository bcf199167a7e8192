import csv, sys, collections
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"][:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
    print(k)
    for c, v in d.items():
        print("   %-28s total %.4g  per-dispatch %.4g  (n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
