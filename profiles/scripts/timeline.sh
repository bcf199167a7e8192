# kernel timeline of one train step: bash profiles/scripts/timeline.sh [precision] [batch]
PREC=${1:-bf16}; BATCH=${2:-520}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/tl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ADN_BENCH_B=$BATCH
timeout 300 rocprofv3 --kernel-trace -d $OUT/tr -o tl --output-format csv -- python3 $ROOT/bench.py --only-train-steps --steps 6 --warmup 3 --precision $PREC > $OUT/tl.log 2>&1
python3 $ROOT/profiles/scripts/timeline.py $(find $OUT/tr -name "tl_kernel_trace.csv" | head -1) > $OUT/timeline_${PREC}_b$BATCH.txt
rm -rf $OUT/tr; cat $OUT/timeline_${PREC}_b$BATCH.txt
