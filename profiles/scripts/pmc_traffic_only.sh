#!/bin/bash
# the two PMC passes + traffic JSON of one arithmetic (collect.sh's recipe, alone):  bash profiles/scripts/pmc_traffic_only.sh <precision> <commit>
set -u
PREC=${1:-bf16x3}; COMMIT=${2:-unknown}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/collect; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --only-train-steps --steps 3 --warmup 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pA_$PREC -o a --output-format csv -- python3 $B --precision $PREC > $OUT/pA_$PREC.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pB_$PREC -o b --output-format csv -- python3 $B --precision $PREC > $OUT/pB_$PREC.log 2>&1
python3 $ROOT/profiles/make_traffic_json.py $(find $OUT/pA_$PREC -name "a_counter_collection.csv" | head -1) $(find $OUT/pB_$PREC -name "b_counter_collection.csv" | head -1) $PREC $COMMIT 4 > $OUT/pmc_traffic_$PREC.json
rm -rf $OUT/pA_$PREC $OUT/pB_$PREC
