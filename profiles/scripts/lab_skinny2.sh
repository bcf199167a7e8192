#!/bin/bash
# A/B of the wide nn kernel (N <= 160, B^T staged in passes) on the first LSTM's input gradient.  (profiles/r05/lab_skinny2.txt also
# holds the K <= 64 kernel's withdrawn variants -- column tiles per workgroup 2 / 4, store pieces 8 / 16 bytes -- and their A/B
# against the kept form on one box.)
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
done
