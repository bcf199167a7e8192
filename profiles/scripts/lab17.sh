export LAB_PAD=64
for mm in 256 128; do echo "=== ADN_GEMM_PP_MIN_M=$mm"; for c in "dW lstm TN" "dW lstm-in"; do ADN_GEMM_PP_MIN_M=$mm LAB_VERIFY=1 ADN_GEMM_PP=4 profiles/gemm_lab "$c" | grep -v "^case"; ADN_GEMM_PP_MIN_M=$mm LAB_GROUPS=3 ADN_GEMM_PP=4 profiles/gemm_lab "$c" | grep -v "^case"; done; done
