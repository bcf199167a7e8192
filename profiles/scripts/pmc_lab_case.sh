#!/bin/bash
# HBM-side counters of ONE gemm_lab case (two PMC passes):  bash profiles/scripts/pmc_lab_case.sh "<case>" [env assignments ...]
# FETCH_SIZE is in 32-byte units on gfx950 after the guide's correction (x2 of the raw 64-byte reading is applied by make_traffic_json;
# here raw per-dispatch values are printed: multiply FETCH_SIZE by 64 x 2 and WRITE_SIZE by 64 for bytes -- see MI355X_MICROARCH.md).
set -u
CASE="$1"; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_lab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/a -o a --output-format csv -- $ROOT/profiles/gemm_lab "$CASE" > $OUT/a.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/b -o b --output-format csv -- $ROOT/profiles/gemm_lab "$CASE" > $OUT/b.log 2>&1
python3 $ROOT/profiles/pmc_summary.py $(find $OUT/a -name "a_counter_collection.csv" | head -1) | grep -A2 "skinny\|gemm_" | head -30
python3 $ROOT/profiles/pmc_summary.py $(find $OUT/b -name "b_counter_collection.csv" | head -1) | grep -A4 "skinny\|gemm_" | head -40
rm -rf $OUT/a $OUT/b
