set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab16; mkdir -p $OUT
export LAB_PAD=64
( echo "=== as shipped"; timeout 200 profiles/gemm_lab 2>&1 | grep -v "^case"
  echo "=== pp forced"; ADN_GEMM_PP=4 timeout 200 profiles/gemm_lab 2>&1 | grep -v "^case"
  echo "=== groups 3 as shipped"; LAB_GROUPS=3 timeout 200 profiles/gemm_lab 2>&1 | grep -v "^case"
  echo "=== groups 3 pp forced"; LAB_GROUPS=3 ADN_GEMM_PP=4 timeout 200 profiles/gemm_lab 2>&1 | grep -v "^case" ) > $OUT/all.txt 2>&1
python3 - <<'PY'
import re, collections
rows = collections.OrderedDict(); cfg = None
for line in open("gpurun_out/lab16/all.txt"):
    if line.startswith("==="): cfg = line.strip("= \n"); continue
    m = re.match(r"(.{27}) (\w\w)\s+(\d+)\s+(\d+)\s+(\d+) \|\s+([\d.]+)", line)
    if m: rows.setdefault(m.group(1).strip() + " %s %s %s %s" % (m.group(2), m.group(3), m.group(4), m.group(5)), {})[cfg] = float(m.group(6))
cfgs = ["as shipped", "pp forced", "groups 3 as shipped", "groups 3 pp forced"]
print("%-50s" % "case", " ".join("%12s" % c[-12:] for c in cfgs))
for k, r in rows.items(): print("%-50s" % k, " ".join("%12.1f" % r.get(c, float("nan")) for c in cfgs))
PY
