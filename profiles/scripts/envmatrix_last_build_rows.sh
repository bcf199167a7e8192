for e in "X=1" "ADN_GEMM_PP=0" "ADN_GEMM_PP=7" "ADN_NO_GROUPED_BACKWARD=1" "ADN_LSTM_NO_CLUSTER=1" "ADN_DETERMINISTIC=1" "ADN_STREAMS=1" "ADN_GEMM_TAIL_SPLIT=1" "ADN_NO_RELU_BITS=1" "ADN_GEMM_PP=8"; do
  echo "=== $e"
  env $e python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_geometry.py tests/test_gpu_last_head.py tests/test_gpu_adenet_v1.py tests/test_gpu_runner.py tests/test_gpu_batch.py tests/test_gpu_compact.py tests/test_gpu_buckets.py -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2
done
# the bf16x3 mode's tests and its fuzz under the switches that change ITS paths
for e in "X=1" "ADN_X3_NO_PLANES=1" "ADN_X3_NO_LEAN=1" "ADN_LSTM_DG_FP32=1" "ADN_X3_NO_GROUPS=1" "ADN_X3_NO_MASK_SHADOWS=1" "ADN_LSTM_NO_X3_CLUSTER_BWD=1" "ADN_LSTM_NO_X3_WIDE=1" "ADN_LSTM_CUS=64" "ADN_GEMM_PP=0" "ADN_NO_GROUPED_BACKWARD=1" "ADN_STREAMS=1" "ADN_X3_MIN_WORK=0" "ADN_DETERMINISTIC=1" "ADN_GEMM_NO_X3F=1" "ADN_GEMM_X3F=tn" "ADN_GEMM_NO_SKINNY=1" "ADN_GEMM_SKINNY_ALL=1" "ADN_GEMM_NO_SKINNY_WIDE=1" "ADN_GEMM_SKINNY_NO_XCD=1" "ADN_MIXED_LSTM_X3=1" "ADN_MIXED_BOTH_PLANES=1" "ADN_FP32_RESIDENT=1" "ADN_GEMM_TAIL_SPLIT=1" "ADN_NO_COMPACT=1" "ADN_BF16_NO_SHADOW=1"; do
  echo "=== bf16x3: $e"
  env $e python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_fuzz.py tests/test_gpu_compact.py tests/test_gpu_buckets.py -q -x -k "x3 or bf16x3 or mixed or plane or skinny or compact or bucket" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2
done
