set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02c; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; tail -c 1500 $OUT/bench_bf16.json
ADN_GEMM_PP=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/bench_bf16_nopp.json 2>/dev/null; tail -c 600 $OUT/bench_bf16_nopp.json
