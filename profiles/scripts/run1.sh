set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02a; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmcM -o m --output-format csv -- python3 $B --precision bf16 > $OUT/pmcM.log 2>&1
python3 $ROOT/profiles/pmc_summary.py $(find $OUT/pmcM -name "m_counter_collection.csv" | head -1) > $OUT/pmc_mfma_bf16.txt
rm -rf $OUT/pmcM
cd $ROOT
tail -3 $OUT/pytest.log; head -40 $OUT/pmc_mfma_bf16.txt
