set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab15; mkdir -p $OUT
export LAB_PAD=64
( for b in 2 1; do echo "=== ADN_GEMM_PP_BARRIERS=$b"; for c in "dW fc2" "x3 fwd fc2"; do ADN_GEMM_PP_BARRIERS=$b ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab_stamps "$c" | grep -v "^case"; done; done ) > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
