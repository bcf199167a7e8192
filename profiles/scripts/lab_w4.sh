# A/B of the four-wave LDS-DMA kernel (ADN_GEMM_PP=7) against the eight-wave one (ADN_GEMM_PP=4) in profiles/gemm_lab
export LAB_PAD=64
cd $GRAFT_REPO_ROOT
for mode in ${MODES:-7}; do
echo "=== ADN_GEMM_PP=$mode verify"
for c in "fwd fc1 bias" "dX fc2 lean y colsum" "dW fc1" "dW fc2" "odd edges" "fwd fc2 lean" "dcat" "dW agg-cat"; do LAB_VERIFY=1 ADN_GEMM_PP=$mode timeout 120 profiles/gemm_lab "$c" 2>&1 | grep -v "^case" | tail -4; done
done
for mode in ${MODES:-7}; do
echo "=== ADN_GEMM_PP=$mode timing"
ADN_GEMM_PP=$mode timeout 200 profiles/gemm_lab 2>&1 | grep -v "narrow\|xproj K\|dX bn\|dW bn\|dW lstm"
echo "=== ADN_GEMM_PP=$mode groups of 3"
LAB_GROUPS=3 ADN_GEMM_PP=$mode timeout 200 profiles/gemm_lab 2>&1 | grep "fwd\|dX fc\|dW fc\|dW agg"
done
