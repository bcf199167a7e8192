set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab4; mkdir -p $OUT
export LAB_PAD=64
( for c in "dW fc2" "dW fc1" "odd edges dW"; do
  echo "== $c: plain mapping"; ADN_GEMM_NO_XCD_SLICES=1 ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab "$c" | grep -v "^case"
  echo "== $c: xcd slices (as selected)"; LAB_VERIFY=1 ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab "$c" | grep -v "^case"
  for sp in 4 8 16; do echo "== $c: splits=$sp"; LAB_VERIFY=1 ADN_GEMM_PP_SPLITS=$sp ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab "$c" | grep -v "^case"; done
done ) > $OUT/tn.txt 2>&1
cat $OUT/tn.txt
