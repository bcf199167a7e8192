set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab6; mkdir -p $OUT
export LAB_PAD=64
( for c in "dW fc2" "x3 fwd fc1" "x3 fwd fc2"; do ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab_stamps "$c"; done ) > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
