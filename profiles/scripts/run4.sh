set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02e; mkdir -p $OUT
python -m pytest tests/test_gpu_bf16x3.py -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --accurate-precision bf16x3 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02e/bench_bf16.json").read().strip().splitlines()[-1])
print("bf16", d["value"], d["ms_per_step"], d["roofline"]["frac"])
a = d.get("accurate")
print("accurate", a["mode"], a["value"], a["ms_per_step"], a["roofline"]["achieved"], a.get("kernel_ms_per_step"))
PY
