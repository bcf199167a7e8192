# ping-pong GEMM kernel: verification + A/B against the register-staged kernels (same process would be better; same box here)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab1; mkdir -p $OUT
export LAB_PAD=64
for f in "fwd fc1" "dX fc2" "fwd fc2" "dW fc1" "dW fc2" "agg-cat" "dcat" "odd edges" "fwd fc3" "dX fc3"; do
  echo "=== $f (ping-pong, verify)"; LAB_VERIFY=1 timeout 120 profiles/gemm_lab "$f" 2>&1 | grep -v "^case"
done > $OUT/verify.txt 2>&1
echo "=== groups=3 verify" >> $OUT/verify.txt
for f in "fwd fc1 bias" "dX fc2 lean y colsum" "fwd fc2" "dW fc1" "dW fc2"; do
  LAB_GROUPS=3 LAB_VERIFY=1 timeout 120 profiles/gemm_lab "$f" 2>&1 | grep -v "^case"
done >> $OUT/verify.txt 2>&1
echo "=== old kernels (ADN_GEMM_PP=0)" > $OUT/ab.txt
ADN_GEMM_PP=0 timeout 200 profiles/gemm_lab >> $OUT/ab.txt 2>&1
echo "=== ping-pong auto" >> $OUT/ab.txt
timeout 200 profiles/gemm_lab >> $OUT/ab.txt 2>&1
for m in 4 5 6; do echo "=== forced mode $m" >> $OUT/ab.txt; ADN_GEMM_PP=$m timeout 200 profiles/gemm_lab >> $OUT/ab.txt 2>&1; done
echo "=== groups 3 auto" >> $OUT/ab.txt
LAB_GROUPS=3 timeout 200 profiles/gemm_lab >> $OUT/ab.txt 2>&1
tail -50 $OUT/verify.txt
