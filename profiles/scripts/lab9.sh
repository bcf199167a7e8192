set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab9; mkdir -p $OUT
export LAB_PAD=64
( for fl in 0 64 56 120; do echo "=== flags $fl (8: no DMA, 16: no fragment reads, 32: no MFMA, 64: no epilogue)"; for c in "dW fc2" "x3 fwd fc2" "x3 fwd fc1"; do ADN_GEMM_PP_FLAGS=$fl ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab "$c" | grep -v "^case"; done; done ) > $OUT/ablate2.txt 2>&1
cat $OUT/ablate2.txt
