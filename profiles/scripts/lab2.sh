# why is the NN 256x256 ping-pong instantiation slow?  counters for one shape in three configurations
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lab2; mkdir -p $OUT
export LAB_PAD=64
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
run() {  # name, env..., case
  name=$1; shift
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    env "$@" timeout 120 rocprofv3 --pmc $pass -d $OUT/p_${name}_$tag -o x --output-format csv -- $ROOT/profiles/gemm_lab "$CASE" > $OUT/log_${name}_$tag.txt 2>&1
    python3 $ROOT/profiles/pmc_summary.py $(find $OUT/p_${name}_$tag -name "x_counter_collection.csv" | head -1) 2>&1 | grep -A9 "pp_kernel" | head -12 >> $OUT/sum_$name.txt
    rm -rf $OUT/p_${name}_$tag
  done
}
CASE="fwd fc1 bias"
run nn4 ADN_GEMM_PP=4 
run nn5 ADN_GEMM_PP=5
CASE="dW fc1"
run tn4 ADN_GEMM_PP=4
cd $ROOT
for f in $OUT/sum_*.txt; do echo "== $f"; cat $f; done
