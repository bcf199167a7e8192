#!/bin/bash
# A/B of the skinny streaming kernels (gemm_skinny.hip) against the register-staged / split-image paths on the train step's narrow
# shapes, plain bf16 and bf16x3 over planes, 1 and 3 problems per launch, with the sampled double-precision check.
export LAB_PAD=64 LAB_VERIFY=1
for planes in 0 1; do
  for groups in 1 3; do
    for off in 1 0; do
      echo "=== planes=$planes groups=$groups ADN_GEMM_NO_SKINNY=$off"
      for c in "narrow fwd bn" "narrow fwd cls" "narrow dX cls" "narrow dW cls" "dX bn lean" "dW bn TN"; do
        if [ $planes = 1 ]; then export LAB_PLANES=1; else unset LAB_PLANES; fi
        if [ $off = 1 ]; then export ADN_GEMM_NO_SKINNY=1; else unset ADN_GEMM_NO_SKINNY; fi
        LAB_GROUPS=$groups timeout 120 profiles/gemm_lab "$c" | grep -v "^case"
      done
    done
  done
done
