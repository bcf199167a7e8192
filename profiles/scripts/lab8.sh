set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab8; mkdir -p $OUT
export LAB_PAD=64
cd /tmp && export TMPDIR=/tmp
for fl in 0 4; do
  for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    ADN_GEMM_PP_FLAGS=$fl ADN_GEMM_PP=4 timeout 120 rocprofv3 --pmc $pass -d $OUT/p -o x --output-format csv -- $ROOT/profiles/gemm_lab "dW fc2" > $OUT/log_${fl}_$tag.txt 2>&1
    echo "== flags $fl $tag" >> $OUT/sum.txt
    python3 $ROOT/profiles/pmc_summary.py $(find $OUT/p -name "x_counter_collection.csv" | head -1) 2>&1 | grep -A4 "pp_kernel" | head -6 >> $OUT/sum.txt
    rm -rf $OUT/p
  done
done
cat $OUT/sum.txt
