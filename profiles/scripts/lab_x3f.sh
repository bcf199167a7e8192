#!/bin/bash
# A/B of the fused-plane kernel (gemm_x3f.hip) against the ping-pong kernel's three K-segments on the train step's GEMM shapes,
# bf16x3 over hi / lo planes, with the sampled double-precision check of every result; then the PLANES = false body
# (ADN_GEMM_PP=8) against the ping-pong kernel in plain bf16.   bash profiles/scripts/lab_x3f.sh > gpurun_out/lab_x3f.txt
export LAB_PAD=64
for groups in 1 3; do
  echo "=== bf16x3 over planes, three K-segments (ADN_GEMM_NO_X3F=1), LAB_GROUPS=$groups"
  LAB_PLANES=1 LAB_VERIFY=1 LAB_GROUPS=$groups ADN_GEMM_NO_X3F=1 timeout 300 profiles/gemm_lab $1
  echo "=== bf16x3 over planes, fused-plane kernel, LAB_GROUPS=$groups"
  LAB_PLANES=1 LAB_VERIFY=1 LAB_GROUPS=$groups timeout 300 profiles/gemm_lab $1
done
echo "=== plain bf16, ping-pong kernel forced (ADN_GEMM_PP=4), LAB_GROUPS=3"
LAB_GROUPS=3 ADN_GEMM_PP=4 timeout 300 profiles/gemm_lab $1
echo "=== plain bf16, 32x32x16 body forced (ADN_GEMM_PP=8), LAB_GROUPS=3"
LAB_VERIFY=1 LAB_GROUPS=3 ADN_GEMM_PP=8 timeout 300 profiles/gemm_lab $1
