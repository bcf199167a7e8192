#!/bin/bash
# Collects the judged artifacts of one build on the GPU box into gpurun_out/collect/ (copy what is kept into profiles/rNN/):
#   bench lines (bf16 incl. the accurate-mode sub-run), rocprofv3 kernel stats, the PMC passes (HBM traffic, L2 hit rate,
#   MFMA-busy cycles) + traffic JSON, per-launch GEMM table, step breakdown, the GEMM lab A/B of the ping-pong kernel,
#   conv AE counters.
# Usage (from the repo root, on an MI355X):   bash profiles/collect.sh [rNN] [commit]
set -u
ROUND=${1:-r06}; COMMIT=${2:-unknown}
LABS=${LABS:-0}        # 1: also the GEMM labs / conv auto-encoder passes (kernels unchanged since round 5: profiles/r05/ holds them)
export BD_INPUTS=bench   # breakdown passes: the batch resident as bench.py hands it over (bfloat16 / hi-lo planes)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/collect
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# counter passes: NOTHING but 1 + 3 train steps of the B=520 workload (no pre-warm, evaluation, B=26 sub-run): every launch the
# counters see is a launch of the headline step (make_traffic_json.py checks the launch counts against PMC_STEPS)
PMC_STEPS=4
B="$ROOT/bench.py --only-train-steps --steps 3 --warmup 1"
# (kernel-stats passes: one per arithmetic mode, the mode under test as --precision)
for prec in bf16 f32 bf16x3 mixed; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks_$prec -o ks --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --accurate-precision none --no-reference-minibatch --precision $prec > $OUT/ks_$prec.log 2>&1
  cp $(find $OUT/ks_$prec -name "ks_kernel_stats.csv" | head -1) $OUT/final_${prec}_kernel_stats.csv
done
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmcA -o a --output-format csv -- python3 $B --precision bf16 > $OUT/pmcA.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmcB -o b --output-format csv -- python3 $B --precision bf16 > $OUT/pmcB.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmcM -o m --output-format csv -- python3 $B --precision bf16 > $OUT/pmcM.log 2>&1
FA=$(find $OUT/pmcA -name "a_counter_collection.csv" | head -1); FB=$(find $OUT/pmcB -name "b_counter_collection.csv" | head -1)
FM=$(find $OUT/pmcM -name "m_counter_collection.csv" | head -1)
python3 $ROOT/profiles/pmc_summary.py $FA > $OUT/pmc_final_bf16_fA.txt
python3 $ROOT/profiles/pmc_summary.py $FB > $OUT/pmc_final_bf16_fB.txt
python3 $ROOT/profiles/pmc_summary.py $FM mfma > $OUT/pmc_mfma_bf16.txt
python3 $ROOT/profiles/make_traffic_json.py $FA $FB bf16 $COMMIT $PMC_STEPS > $OUT/pmc_traffic_bf16.json
# the same two passes for the bf16x3 mode (roofline.traffic of the bench line's `accurate` sub-object)
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmcA3 -o a --output-format csv -- python3 $B --precision bf16x3 > $OUT/pmcA3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmcB3 -o b --output-format csv -- python3 $B --precision bf16x3 > $OUT/pmcB3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmcM3 -o m --output-format csv -- python3 $B --precision bf16x3 > $OUT/pmcM3.log 2>&1
python3 $ROOT/profiles/pmc_summary.py $(find $OUT/pmcM3 -name "m_counter_collection.csv" | head -1) mfma > $OUT/pmc_mfma_bf16x3.txt
python3 $ROOT/profiles/make_traffic_json.py $(find $OUT/pmcA3 -name "a_counter_collection.csv" | head -1) $(find $OUT/pmcB3 -name "b_counter_collection.csv" | head -1) bf16x3 $COMMIT $PMC_STEPS > $OUT/pmc_traffic_bf16x3.json
# round 5: the f32 mode's and the mixed mode's counter traffic too (roofline.traffic of `accurate_f32` / `mixed`)
for prec in f32 mixed; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmcA_$prec -o a --output-format csv -- python3 $B --precision $prec > $OUT/pmcA_$prec.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmcB_$prec -o b --output-format csv -- python3 $B --precision $prec > $OUT/pmcB_$prec.log 2>&1
  python3 $ROOT/profiles/make_traffic_json.py $(find $OUT/pmcA_$prec -name "a_counter_collection.csv" | head -1) $(find $OUT/pmcB_$prec -name "b_counter_collection.csv" | head -1) $prec $COMMIT $PMC_STEPS > $OUT/pmc_traffic_$prec.json
  rm -rf $OUT/pmcA_$prec $OUT/pmcB_$prec
done
ADN_GEMM_TRACE=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/bd -o bd --output-format csv -- python3 $ROOT/profiles/gemm_breakdown.py run 2> $OUT/gemm_trace.txt > $OUT/bd.log
BD=$(find $OUT/bd -name "bd_kernel_trace.csv" | head -1)
python3 $ROOT/profiles/gemm_breakdown.py join $OUT/gemm_trace.txt $BD > $OUT/gemm_breakdown_bf16.txt
python3 $ROOT/profiles/step_breakdown.py $BD > $OUT/step_breakdown_bf16.txt
# the same two tables for the bf16x3 (fp32-grade) mode
ADN_PRECISION=bf16x3 ADN_GEMM_TRACE=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/bd3 -o bd --output-format csv -- python3 $ROOT/profiles/gemm_breakdown.py run 2> $OUT/gemm_trace_x3.txt > $OUT/bd3.log
BD3=$(find $OUT/bd3 -name "bd_kernel_trace.csv" | head -1)
python3 $ROOT/profiles/gemm_breakdown.py join $OUT/gemm_trace_x3.txt $BD3 > $OUT/gemm_breakdown_bf16x3.txt
python3 $ROOT/profiles/step_breakdown.py $BD3 > $OUT/step_breakdown_bf16x3.txt
# ... and for the mixed mode (bf16x3 forward, bf16 backward products)
ADN_PRECISION=mixed ADN_GEMM_TRACE=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/bdm -o bd --output-format csv -- python3 $ROOT/profiles/gemm_breakdown.py run 2> $OUT/gemm_trace_mixed.txt > $OUT/bdm.log
BDM=$(find $OUT/bdm -name "bd_kernel_trace.csv" | head -1)
python3 $ROOT/profiles/gemm_breakdown.py join $OUT/gemm_trace_mixed.txt $BDM > $OUT/gemm_breakdown_mixed.txt
python3 $ROOT/profiles/step_breakdown.py $BDM > $OUT/step_breakdown_mixed.txt
# ... the deterministic schedule of the bf16 step, and the bf16 step at B = 65 (the per-rank batch of an 8-way strong-scaled run)
ADN_DETERMINISTIC=1 ADN_GEMM_TRACE=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/bdd -o bd --output-format csv -- python3 $ROOT/profiles/gemm_breakdown.py run 2> /dev/null > $OUT/bdd.log
python3 $ROOT/profiles/step_breakdown.py $(find $OUT/bdd -name "bd_kernel_trace.csv" | head -1) > $OUT/step_breakdown_bf16_deterministic.txt
BD_BATCH=65 ADN_GEMM_TRACE=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/bd65 -o bd --output-format csv -- python3 $ROOT/profiles/gemm_breakdown.py run 2> $OUT/gemm_trace_b65.txt > $OUT/bd65.log
python3 $ROOT/profiles/gemm_breakdown.py join $OUT/gemm_trace_b65.txt $(find $OUT/bd65 -name "bd_kernel_trace.csv" | head -1) > $OUT/gemm_breakdown_bf16_B65.txt
python3 $ROOT/profiles/step_breakdown.py $(find $OUT/bd65 -name "bd_kernel_trace.csv" | head -1) > $OUT/step_breakdown_bf16_B65.txt
rm -rf $OUT/bdm $OUT/bdd $OUT/bd65 $OUT/gemm_trace_mixed.txt $OUT/gemm_trace_b65.txt
if [ "$LABS" = 1 ]; then
# conv auto-encoder: HBM bytes of its train step at batch 1024 (two PMC passes; 10 steps in the process)
( cd /tmp; CAE_BATCH=1024 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/caeA -o a --output-format csv -- python3 $ROOT/profiles/convae_profile.py bf16 > $OUT/caeA.log 2>&1
  CAE_BATCH=1024 timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/caeB -o b --output-format csv -- python3 $ROOT/profiles/convae_profile.py bf16 > $OUT/caeB.log 2>&1
  python3 $ROOT/profiles/pmc_bytes_total.py $(find $OUT/caeA -name "a_counter_collection.csv" | head -1) $(find $OUT/caeB -name "b_counter_collection.csv" | head -1) 10 > $OUT/pmc_bytes_convae_b1024.txt; rm -rf $OUT/caeA $OUT/caeB )
# conv auto-encoder: MFMA-busy share of its GEMM kernels
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmcC -o c --output-format csv -- python3 $ROOT/profiles/convae_profile.py > $OUT/pmcC.log 2>&1
python3 $ROOT/profiles/pmc_summary.py $(find $OUT/pmcC -name "c_counter_collection.csv" | head -1) mfma > $OUT/pmc_mfma_convae.txt
fi
cd $ROOT
if [ "$LABS" = 1 ]; then
# GEMM lab: the ping-pong kernel (forced 256 x 256 tiles, auto selection, 3 grouped problems) against the register-staged kernels
( export LAB_PAD=64
  echo "=== register-staged kernels (ADN_GEMM_PP=0)"; ADN_GEMM_PP=0 timeout 200 profiles/gemm_lab
  echo "=== selection as shipped"; timeout 200 profiles/gemm_lab
  echo "=== ping-pong forced, 256 x 256 tiles (ADN_GEMM_PP=4)"; ADN_GEMM_PP=4 timeout 200 profiles/gemm_lab
  echo "=== three problems per launch (LAB_GROUPS=3), selection as shipped"; LAB_GROUPS=3 timeout 200 profiles/gemm_lab
  echo "=== three problems per launch, ping-pong forced"; LAB_GROUPS=3 ADN_GEMM_PP=4 timeout 200 profiles/gemm_lab ) > $OUT/gemm_lab_pp.txt 2>&1
# round 5: the fused-plane kernel against the three K-segments (+ the 32x32x16 body in plain bf16), in-kernel stamps and clock, the
# skinny kernels against the register-staged / split-image paths
timeout 900 bash profiles/scripts/lab_x3f.sh > $OUT/lab_x3f.txt 2>&1
timeout 300 bash profiles/scripts/x3f_stamps.sh > $OUT/x3f_stamps.txt 2>&1
timeout 600 bash profiles/scripts/lab_skinny.sh > $OUT/lab_skinny.txt 2>&1
timeout 600 bash profiles/scripts/lab_skinny2.sh > $OUT/lab_skinny2.txt 2>&1
fi
timeout 600 python3 profiles/scripts/strong_scaling_forecast.py bf16 > $OUT/strong_scaling_forecast.txt 2>&1
if [ "$LABS" = 1 ]; then
timeout 200 python3 profiles/hipblaslt_calibration.py > $OUT/hipblaslt_calibration.txt 2>/dev/null
timeout 300 python3 profiles/convae_bench.py > $OUT/convae_bench.txt 2>/dev/null
fi
# the bench lines last, so that roofline.traffic comes from the PMC passes of THIS build
mkdir -p profiles/$ROUND && cp $OUT/pmc_traffic_bf16.json profiles/$ROUND/pmc_traffic_bf16.json && cp $OUT/pmc_traffic_bf16x3.json profiles/$ROUND/pmc_traffic_bf16x3.json
cp $OUT/pmc_traffic_f32.json $OUT/pmc_traffic_mixed.json profiles/$ROUND/ 2>/dev/null
# data parallel machinery on one GPU (no transfer): bucket events / per-bucket Adam / grouped collectives, and the HIP-graph A/B
# of the reference minibatch
timeout 300 python3 profiles/scripts/dp_order_bench.py > $OUT/dp_order_bench.txt 2>&1
timeout 600 bash profiles/scripts/dp_forced.sh > $OUT/dp_forced.txt 2>&1
timeout 600 python3 profiles/configs_bench.py > $OUT/configs_bench.txt 2>/dev/null
# round 4: the epoch through the product entry point (runners/nstream.fit, HBM-resident splits), the deterministic mode's cost,
# the folded input projection A/B, the conv auto-encoder's kernel table at batch 1024
timeout 600 python3 profiles/epoch_bench.py > $OUT/epoch_bench.txt 2>/dev/null
( for d in 0 1; do echo -n "ADN_DETERMINISTIC=$d: "; ADN_DETERMINISTIC=$d timeout 300 python3 bench.py --no-cpu-baseline --accurate-precision all --no-runner --no-profile 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("bf16 %.3f ms/step, bf16x3 %.3f, mixed %.3f, f32 %.3f, B=26 %.3f" % (d["ms_per_step"], d["accurate"]["ms_per_step"], d["mixed"]["ms_per_step"], d["accurate_f32"]["ms_per_step"], d["reference_minibatch"]["ms_per_step"]))'; done
  for v in "ADN_NO_COMPACT=1" "ADN_GEMM_TAIL_SPLIT=1"; do echo -n "$v: "; env $v timeout 300 python3 bench.py --no-cpu-baseline --accurate-precision all --no-runner --no-profile --no-reference-minibatch 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("bf16 %.3f ms/step, bf16x3 %.3f, mixed %.3f" % (d["ms_per_step"], d["accurate"]["ms_per_step"], d["mixed"]["ms_per_step"]))'; done
) > $OUT/mode_costs.txt 2>&1
# the gather kernel's own time inside the runner (B = 26 and B = 520 launches in one table), and the CLI driver end to end
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/eb -o eb --output-format csv -- python3 $ROOT/profiles/epoch_bench.py --precisions bf16 --epochs 4 > $OUT/eb.log 2>&1; grep -i "batch_gather\|\"Name\"" $(find $OUT/eb -name "eb_kernel_stats.csv" | head -1) > $OUT/batch_gather_kernel_stats.csv; rm -rf $OUT/eb )
timeout 600 python3 profiles/scripts/runner_demo.py bf16x3 12 > $OUT/runner_demo.txt 2>&1
if [ "$LABS" = 1 ]; then
ADN_GEMM_TRACE=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/cbd -o bd --output-format csv -- python3 $ROOT/profiles/scripts/convae_gemm_breakdown.py run 2> $OUT/cae_trace.txt > $OUT/cbd.log
python3 $ROOT/profiles/scripts/convae_gemm_breakdown.py join $OUT/cae_trace.txt $(find $OUT/cbd -name "bd_kernel_trace.csv" | head -1) > $OUT/convae_gemm_breakdown.txt; rm -rf $OUT/cbd
( cd /tmp; CAE_BATCH=1024 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/cae -o cae --output-format csv -- python3 $ROOT/profiles/convae_profile.py bf16 > $OUT/cae.log 2>&1; cp $(find $OUT/cae -name "cae_kernel_stats.csv" | head -1) $OUT/convae_bf16_b1024_kernel_stats.csv; rm -rf $OUT/cae )
fi
for cfg in "bf16 520" "bf16x3 520" "mixed 520" "bf16 26"; do set -- $cfg; bash profiles/scripts/timeline.sh $1 $2 > /dev/null 2>&1; cp gpurun_out/tl/timeline_$1_b$2.txt $OUT/; done
timeout 400 python3 bench.py --precision bf16 2>/dev/null | tail -1 > $OUT/final_bf16_bench.json
rm -rf $OUT/ks_bf16 $OUT/ks_f32 $OUT/ks_bf16x3 $OUT/pmcA $OUT/pmcB $OUT/pmcA3 $OUT/pmcB3 $OUT/pmcM3 $OUT/pmcM $OUT/pmcC $OUT/bd $OUT/bd3 $OUT/gemm_trace_x3.txt
ls -la $OUT
