set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab10; mkdir -p $OUT
export LAB_PAD=64
( for c in "fwd fc1" "dX fc2" "fwd fc2" "dW fc1" "dW fc2" "odd edges" "x3 fwd fc2" "x3 fwd fc1" "x3 dX fc2"; do
    echo "=== $c: pp forced, verify"; LAB_VERIFY=1 ADN_GEMM_PP=4 timeout 120 profiles/gemm_lab "$c" 2>&1 | grep -v "^case"
  done
  echo "=== groups 3, pp forced, verify"; LAB_GROUPS=3 LAB_VERIFY=1 ADN_GEMM_PP=4 timeout 200 profiles/gemm_lab 2>&1 | grep -v "^case"
  echo "=== register-staged (PP=0)"; ADN_GEMM_PP=0 timeout 200 profiles/gemm_lab 2>&1 | grep -v "^case"
  echo "=== as shipped"; timeout 200 profiles/gemm_lab 2>&1 | grep -v "^case"
  for m in 5 6; do echo "=== forced mode $m"; LAB_VERIFY=1 ADN_GEMM_PP=$m timeout 200 profiles/gemm_lab 2>&1 | grep -v "^case"; done
) > $OUT/ab.txt 2>&1
( for c in "dW fc2" "x3 fwd fc1" "x3 fwd fc2"; do ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab_stamps "$c"; done ) > $OUT/stamps.txt 2>&1
tail -5 $OUT/stamps.txt
