set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab3; mkdir -p $OUT
export LAB_PAD=64
for f in "fwd fc1" "dX fc2" "fwd fc2" "dcat" "odd edges" "dX fc3 lean y colsum"; do
  echo "=== $f (ping-pong, verify)"; ADN_GEMM_PP=4 LAB_VERIFY=1 timeout 120 profiles/gemm_lab "$f" 2>&1 | grep -v "^case"
done > $OUT/verify.txt 2>&1
LAB_GROUPS=3 ADN_GEMM_PP=4 LAB_VERIFY=1 timeout 120 profiles/gemm_lab "dX fc2 lean y colsum" >> $OUT/verify.txt 2>&1
for m in 4 5 6; do echo "=== forced mode $m"; ADN_GEMM_PP=$m timeout 200 profiles/gemm_lab; done > $OUT/ab.txt 2>&1
echo "=== groups 3 mode 4" >> $OUT/ab.txt
LAB_GROUPS=3 ADN_GEMM_PP=4 timeout 200 profiles/gemm_lab >> $OUT/ab.txt 2>&1
echo "=== groups 3 mode 5" >> $OUT/ab.txt
LAB_GROUPS=3 ADN_GEMM_PP=5 timeout 200 profiles/gemm_lab >> $OUT/ab.txt 2>&1
grep -c "0/6000" $OUT/verify.txt; grep MISMATCH $OUT/verify.txt | head
