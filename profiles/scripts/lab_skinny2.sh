#!/bin/bash
# A/B of the round's later skinny forms: the wide nn kernel (N <= 160, B^T staged in passes) on the first LSTM's input gradient,
# and the K <= 64 kernel's column tiles per workgroup (2 / 4) and store pieces (8 / 16 bytes).
export LAB_PAD=64 LAB_VERIFY=1
for planes in 0 1; do
  if [ $planes = 1 ]; then export LAB_PLANES=1; else unset LAB_PLANES; fi
  for groups in 1 2; do
    for off in 1 0; do
      if [ $off = 1 ]; then export ADN_GEMM_NO_SKINNY_WIDE=1; else unset ADN_GEMM_NO_SKINNY_WIDE; fi
      echo "=== planes=$planes groups=$groups ADN_GEMM_NO_SKINNY_WIDE=$off"
      LAB_GROUPS=$groups timeout 120 profiles/gemm_lab "narrow dfeat" | grep -v "^case"
      if [ $planes = 0 ] && [ $groups = 1 ]; then timeout 120 profiles/gemm_lab "narrow cae conv3" | grep -v "^case"; fi
    done
  done
  unset ADN_GEMM_NO_SKINNY_WIDE
  for v in "TC=2 ST8" "TC=2" "TC=4 ST8" "TC=4"; do
    unset ADN_GEMM_SKINNY_ST8
    case "$v" in *ST8*) export ADN_GEMM_SKINNY_ST8=1;; esac
    case "$v" in TC=2*) export ADN_GEMM_SKINNY_TC=2;; *) export ADN_GEMM_SKINNY_TC=4;; esac
    echo "=== planes=$planes groups=3 K<=64 kernel: $v"
    for c in "dX bn lean" "narrow dX cls"; do
      ADN_GEMM_SKINNY_ALL=1 LAB_GROUPS=3 timeout 120 profiles/gemm_lab "$c" | grep -v "^case"
    done
  done
  unset ADN_GEMM_SKINNY_ST8 ADN_GEMM_SKINNY_TC
done
