#!/bin/bash
# Collects the judged artifacts of one build on the GPU box into gpurun_out/collect/ (copy what is kept into profiles/rNN/):
#   bench lines (bf16, f32), rocprofv3 kernel stats, the two PMC passes + traffic JSON, per-launch GEMM table, step breakdown.
# Usage (from the repo root, on an MI355X):   bash profiles/collect.sh
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/collect
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile"
for prec in bf16 f32; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks_$prec -o ks --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision $prec > $OUT/ks_$prec.log 2>&1
  cp $(find $OUT/ks_$prec -name "ks_kernel_stats.csv" | head -1) $OUT/final_${prec}_kernel_stats.csv
done
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmcA -o a --output-format csv -- python3 $B --precision bf16 > $OUT/pmcA.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmcB -o b --output-format csv -- python3 $B --precision bf16 > $OUT/pmcB.log 2>&1
FA=$(find $OUT/pmcA -name "a_counter_collection.csv" | head -1); FB=$(find $OUT/pmcB -name "b_counter_collection.csv" | head -1)
python3 $ROOT/profiles/pmc_summary.py $FA > $OUT/pmc_final_bf16_fA.txt
python3 $ROOT/profiles/pmc_summary.py $FB > $OUT/pmc_final_bf16_fB.txt
python3 $ROOT/profiles/make_traffic_json.py $FA $FB bf16 > $OUT/pmc_traffic_bf16.json
ADN_GEMM_TRACE=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/bd -o bd --output-format csv -- python3 $ROOT/profiles/gemm_breakdown.py run 2> $OUT/gemm_trace.txt > $OUT/bd.log
BD=$(find $OUT/bd -name "bd_kernel_trace.csv" | head -1)
python3 $ROOT/profiles/gemm_breakdown.py join $OUT/gemm_trace.txt $BD > $OUT/gemm_breakdown_bf16.txt
python3 $ROOT/profiles/step_breakdown.py $BD > $OUT/step_breakdown_bf16.txt
cd $ROOT
# the bench lines last, so that roofline.traffic comes from the PMC passes of THIS build
mkdir -p profiles/r01 && cp $OUT/pmc_traffic_bf16.json profiles/r01/pmc_traffic_bf16.json
timeout 300 python3 bench.py --precision bf16 2>/dev/null | tail -1 > $OUT/final_bf16_bench.json
timeout 600 python3 bench.py --precision f32 2>/dev/null | tail -1 > $OUT/final_f32_bench.json
rm -rf $OUT/ks_bf16 $OUT/ks_f32 $OUT/pmcA $OUT/pmcB $OUT/bd
ls -la $OUT
