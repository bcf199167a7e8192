set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab13; mkdir -p $OUT
export LAB_PAD=64
( for t in 0 256; do for sp in 0 256 512; do echo "=== ADN_GEMM_TILE=$t SPLIT_TARGET=$sp"
    for c in "dW lstm TN" "dW lstm-in" "dW fc3" "dW bn TN" "dW agg-cat"; do
      if [ $t = 0 ]; then T="X=1"; else T="ADN_GEMM_TILE=$t"; fi
      if [ $sp = 0 ]; then S="Y=1"; else S="ADN_GEMM_SPLIT_TARGET=$sp"; fi
      env $T $S LAB_VERIFY=1 ADN_GEMM_PP=0 timeout 100 profiles/gemm_lab "$c" 2>&1 | grep -v "^case"
    done; done; done ) > $OUT/tall_tn.txt 2>&1
grep -v "0/6000" $OUT/tall_tn.txt
