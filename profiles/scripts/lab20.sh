export LAB_PAD=64
echo "=== pp forced, single"; for c in "dX fc2" "dX fc3" "x3 dX"; do LAB_VERIFY=1 ADN_GEMM_PP=4 timeout 120 profiles/gemm_lab "$c" 2>&1 | grep -v "^case"; done
echo "=== groups 3 pp forced"; LAB_GROUPS=3 LAB_VERIFY=1 ADN_GEMM_PP=4 timeout 200 profiles/gemm_lab "dX fc" 2>&1 | grep -v "^case"
echo "=== groups 3 reg-staged"; LAB_GROUPS=3 ADN_GEMM_PP=0 timeout 200 profiles/gemm_lab "dX fc" 2>&1 | grep -v "^case"
