#!/bin/bash
# The data-parallel step on ONE GPU through a single-rank RCCL group (bench.py, ADN_BENCH_FORCE_DP=1): what the bucket
# machinery + the collective launches cost a step before any transfer.  bash profiles/scripts/dp_forced.sh
run() {
  echo -n "$1: "
  env $2 ADN_BENCH_FORCE_DP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 40 --warmup 10 --no-profile --accurate-precision none --no-reference-minibatch --no-cpu-baseline --no-runner 2>/dev/null \
    | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.3f ms/step" % d["ms_per_step"])'
}
echo -n "plain step (no data-parallel machinery): "; python bench.py --steps 40 --warmup 10 --no-profile --accurate-precision none --no-reference-minibatch --no-cpu-baseline --no-runner 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.3f ms/step" % d["ms_per_step"])'
run "one all-reduce of the whole buffer, Adam (no overlap)" "ADN_DP_NO_OVERLAP=1"
run "16 buckets, one collective + event each, Adam x2     " "ADN_DP_NO_COALESCE=1"
run "16 buckets, 6 grouped collectives, Adam x2 (round 3) " "ADN_DP_FINE_BUCKETS=1"
run "16 buckets, 2 grouped collectives, Adam x2 (default) " "X=1"
run "16 buckets, grouped collectives, whole-buffer Adam   " "ADN_DP_WHOLE_BUFFER_ADAM=1"
run "stream-major, one collective per bucket              " "ADN_DP_STREAM_MAJOR=1"
