set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab14; mkdir -p $OUT
export LAB_PAD=64
( for b in 2 1; do echo "=== ADN_GEMM_PP_BARRIERS=$b, pp forced, verify"
    for c in "fwd fc1" "fwd fc2" "dX fc2 lean y colsum" "dW fc1" "dW fc2" "odd edges" "x3 fwd fc2" "x3 fwd fc1"; do
      ADN_GEMM_PP_BARRIERS=$b LAB_VERIFY=1 ADN_GEMM_PP=4 timeout 120 profiles/gemm_lab "$c" 2>&1 | grep -v "^case"
    done
    echo "=== ADN_GEMM_PP_BARRIERS=$b groups 3"; ADN_GEMM_PP_BARRIERS=$b LAB_GROUPS=3 LAB_VERIFY=1 ADN_GEMM_PP=4 timeout 200 profiles/gemm_lab "fwd fc" 2>&1 | grep -v "^case"
  done ) > $OUT/barriers.txt 2>&1
grep -v "0/6000" $OUT/barriers.txt
