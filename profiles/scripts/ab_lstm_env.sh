# A/B of one environment switch on the LSTM kernels, one box, alternating runs: bash profiles/scripts/ab_lstm_env.sh VAR [rounds]
# prints the bench line's LSTM roofline fractions (bf16 headline and bf16x3) and kernel ms per step for VAR unset / VAR=1
VAR=$1; ROUNDS=${2:-2}
for i in $(seq $ROUNDS); do
  for v in 0 1; do
    if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-runner --no-reference-minibatch --accurate-precision bf16x3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; a = d['accurate']['kernel_ms_per_step']; k = d['kernel_ms_per_step']
print('$VAR=$v  bf16 %.3f ms (lstm fwd %.3f bwd %.3f ms; frac %.3f / %.3f) | bf16x3 %.3f ms (lstm fwd %.3f bwd %.3f ms; frac %.3f / %.3f)' % (
    d['ms_per_step'], k['lstm_fwd_step'], k['lstm_bwd_step'], r['lstm_fwd_frac'], r['lstm_bwd_frac'],
    d['config']['parity_grade_ms'], a['lstm_fwd_step'], a['lstm_bwd_step'], r['lstm_fwd_frac_parity_grade'], r['lstm_bwd_frac_parity_grade']))"
  done
done
