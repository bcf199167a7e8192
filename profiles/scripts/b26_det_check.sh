# B = 26 step time under a list of environment settings: bash profiles/scripts/b26_det_check.sh "A=1 B=1" "C=1" ...
for v in "$@"; do
  echo -n "$v: "
  env $v python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-runner --accurate-precision none 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("B=520 %.3f ms | B=26 %.3f ms" % (d["ms_per_step"], d["config"]["reference_minibatch_B26_ms"]))'
done
