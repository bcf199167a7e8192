export LAB_PAD=64 LAB_GROUPS=3
for c in "fwd fc1 bias" "dX fc2 lean y colsum" "dW fc1" "fwd fc2 lean"; do
  echo "=== fused planes: $c"; LAB_PLANES=1 timeout 120 profiles/gemm_lab_stamps "$c"
  echo "=== 3-seg planes: $c"; LAB_PLANES=1 ADN_GEMM_NO_X3F=1 timeout 120 profiles/gemm_lab_stamps "$c"
  echo "=== bf16 PP=8: $c"; ADN_GEMM_PP=8 timeout 120 profiles/gemm_lab_stamps "$c"
  echo "=== bf16 PP=4: $c"; ADN_GEMM_PP=4 timeout 120 profiles/gemm_lab_stamps "$c"
done
