export LAB_PAD=64
for m in 4 5; do echo "=== pp mode $m, min N 96"; ADN_GEMM_PP_MIN_N=96 LAB_VERIFY=1 ADN_GEMM_PP=$m timeout 120 profiles/gemm_lab "narrow cae" 2>&1 | grep -v "^case"; done
echo "=== shipped"; timeout 120 profiles/gemm_lab "narrow cae" 2>&1 | grep -v "^case"
