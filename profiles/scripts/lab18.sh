export LAB_PAD=64
for bin in gemm_lab gemm_lab_alt gemm_lab gemm_lab_alt; do echo "=== $bin"
  for c in "fwd fc1 bias" "fwd fc2" "dX fc2 lean y colsum" "dW fc1" "dW fc2" "odd edges" "x3 fwd fc2"; do LAB_VERIFY=1 ADN_GEMM_PP=4 timeout 120 profiles/$bin "$c" 2>&1 | grep -v "^case\|0/6000"; done
  LAB_GROUPS=3 ADN_GEMM_PP=4 timeout 200 profiles/$bin "fwd fc" 2>&1 | grep -v "^case"
done
