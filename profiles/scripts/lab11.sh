set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab11; mkdir -p $OUT
export LAB_PAD=64
( for t in 0 64 128; do for sp in 0 256 512 1024; do
    echo "=== ADN_GEMM_TILE=$t ADN_GEMM_SPLIT_TARGET=$sp"
    for c in "dW lstm TN" "dW lstm-in" "dW fc3" "dW agg-cat" "dW bn TN" "dX fc3 lean y colsum" "xproj K=150" "fwd fc3"; do
      if [ $t = 0 ]; then T=""; else T="ADN_GEMM_TILE=$t"; fi
      if [ $sp = 0 ]; then S=""; else S="ADN_GEMM_SPLIT_TARGET=$sp"; fi
      env $T $S ADN_GEMM_PP=0 timeout 60 profiles/gemm_lab "$c" 2>&1 | grep -v "^case"
    done; done; done ) > $OUT/tiles.txt 2>&1
tail -5 $OUT/tiles.txt
