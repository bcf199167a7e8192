# the rows of envmatrix.sh that failed in round 6's first pass (test expectations under diagnostic switches), re-run after the fixes
for e in "ADN_GEMM_PP=0"; do
  echo "=== $e"
  env $e python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_geometry.py tests/test_gpu_last_head.py tests/test_gpu_adenet_v1.py tests/test_gpu_runner.py tests/test_gpu_batch.py tests/test_gpu_compact.py -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
done
for e in "ADN_X3_NO_PLANES=1" "ADN_X3_MIN_WORK=0" "ADN_BF16_NO_SHADOW=1"; do
  echo "=== bf16x3: $e"
  env $e python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_fuzz.py tests/test_gpu_compact.py -q -k "x3 or bf16x3 or mixed or plane or skinny or compact" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
done
