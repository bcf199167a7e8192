set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab12; mkdir -p $OUT
export LAB_PAD=64
( for t in 0 64 128 256; do echo "=== ADN_GEMM_TILE=$t"
    for c in "narrow" "xproj K=150" "dX bn lean" "fwd fc3" "dX fc3 lean y colsum"; do
      if [ $t = 0 ]; then T="X=1"; else T="ADN_GEMM_TILE=$t"; fi
      env $T LAB_VERIFY=1 ADN_GEMM_PP=0 timeout 100 profiles/gemm_lab "$c" 2>&1 | grep -v "^case"
    done; done ) > $OUT/tall.txt 2>&1
cat $OUT/tall.txt
