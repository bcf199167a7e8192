# per-launch GEMM table + per-kernel step breakdown of ONE train step; usage: bash profiles/scripts/bd.sh [precision] [batch]
set -u
PREC=${1:-bf16}; BATCH=${2:-520}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/bd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ADN_PRECISION=$PREC BD_BATCH=$BATCH BD_WARM=${BD_WARM:-3} ADN_GEMM_TRACE=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/bd -o bd --output-format csv -- python3 $ROOT/profiles/gemm_breakdown.py run 2> $OUT/gemm_trace.txt > $OUT/bd.log
BD=$(find $OUT/bd -name "bd_kernel_trace.csv" | head -1)
python3 $ROOT/profiles/gemm_breakdown.py join $OUT/gemm_trace.txt $BD > $OUT/gemm_breakdown_${PREC}_b$BATCH.txt
python3 $ROOT/profiles/step_breakdown.py $BD > $OUT/step_breakdown_${PREC}_b$BATCH.txt
rm -rf $OUT/bd
cat $OUT/gemm_breakdown_${PREC}_b$BATCH.txt; head -24 $OUT/step_breakdown_${PREC}_b$BATCH.txt
