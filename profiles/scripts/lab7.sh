set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab7; mkdir -p $OUT
export LAB_PAD=64
( for fl in 0 1 2 3; do echo "=== flags $fl"; for c in "dW fc2" "dW fc1" "x3 fwd fc1" "x3 fwd fc2"; do ADN_GEMM_PP_FLAGS=$fl ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab "$c" | grep -v "^case"; done; done ) > $OUT/prio.txt 2>&1
cat $OUT/prio.txt
