set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/cae; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CAE_BATCH=1024 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks --output-format csv -- python3 $ROOT/profiles/convae_profile.py bf16 > $OUT/ks.log 2>&1
cp $(find $OUT/ks -name "ks_kernel_stats.csv" | head -1) $OUT/cae_kernel_stats.csv
rm -rf $OUT/ks
cd $ROOT
python profiles/convae_bench.py > $OUT/convae_bench.txt 2>&1
head -30 $OUT/cae_kernel_stats.csv | cut -c1-160
cat $OUT/convae_bench.txt
