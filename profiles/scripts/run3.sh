set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02i; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $OUT/pytest.log | tail -8
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02i/bench_bf16.json").read().strip().splitlines()[-1])
print("bf16", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("accurate", "accurate_f32"):
    a = d.get(k)
    if a: print(k, a["mode"], a["value"], a["ms_per_step"], a["roofline"]["achieved"], a["roofline"]["share_of_step"], a.get("kernel_ms_per_step"))
PY
