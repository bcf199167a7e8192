# A/B of one environment switch on ONE box, alternating runs: bash profiles/scripts/ab_env.sh VAR [rounds] [bench precision list]
# prints ms per step of the bf16 headline, bf16x3 and mixed for VAR unset / VAR=1
VAR=$1; ROUNDS=${2:-2}
for i in $(seq $ROUNDS); do
  for v in 0 1; do
    if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
    python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-runner --no-reference-minibatch --accurate-precision ${AB_ACC:-all} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']
print('$VAR=$v  bf16 %.3f ms/step | bf16x3 %s | mixed %s' % (d['ms_per_step'], c.get('parity_grade_ms'), c.get('mixed_ms')))"
  done
done
