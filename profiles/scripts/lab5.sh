set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab5; mkdir -p $OUT
export LAB_PAD=64
( for c in "dX fc2" "dX fc3" "fwd fc1 bias"; do
  echo "== $c pp forced (groups 1)"; LAB_VERIFY=1 ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab "$c" | grep -v "^case"
  echo "== $c pp forced (groups 3)"; LAB_GROUPS=3 ADN_GEMM_PP=4 timeout 100 profiles/gemm_lab "$c" | grep -v "^case"
  echo "== $c old"; ADN_GEMM_PP=0 timeout 100 profiles/gemm_lab "$c" | grep -v "^case"
done ) > $OUT/epi.txt 2>&1
cat $OUT/epi.txt
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
