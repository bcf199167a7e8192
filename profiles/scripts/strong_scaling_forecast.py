"""Strong-scaling forecast from single-GPU measurements (no multi-GPU node is available to the builder): the step of the bench
model at the per-rank batches a strong-scaled AVLetters whole-train batch (520 utterances) leaves -- 520 / 260 / 130 / 65 -- timed
on ONE MI355X, plus a stated model of the 72 MB gradient all-reduce.  Prints the table DESIGN.md 7 quotes.

    python profiles/scripts/strong_scaling_forecast.py [bf16|bf16x3|mixed]

All-reduce model (assumptions, not measurements): ring all-reduce over xGMI moves 2 (N - 1) / N x bytes per rank at the bus
bandwidth RCCL reaches on 8 fully connected MI300-class GPUs (taken as 300 GB/s at N = 8, 200 at N = 4, 100 at N = 2: one / three /
seven of the 7 x 153 GB/s links in use per direction, ~65 % efficiency) + 30 us of launch latency per collective (two per step);
the first collective (43 MB) runs under the layer-0 weight-gradient launch and is taken as hidden up to that launch's length
(scaled with the batch), the second (29 MB) is exposed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
BUS = {2: 100e9, 4: 200e9, 8: 300e9}
HEAD, TAIL, LAT = 43e6, 29e6, 30e-6


def step_ms(batch):
    env = dict(os.environ, ADN_BENCH_B=str(batch))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--precision", prec, "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline", "--accurate-precision", "none", "--no-runner", "--no-reference-minibatch", "--no-profile"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [l for l in res.stdout.strip().splitlines() if l.startswith("{")]
    if not lines:
        raise SystemExit("bench.py failed at B = %d:\n%s" % (batch, res.stderr[-2000:]))
    return json.loads(lines[-1])["ms_per_step"]


t = {b: step_ms(b) for b in (520, 260, 130, 65)}
print("single-GPU step of the bench model, %s: " % prec + ", ".join("B = %d: %.3f ms" % (b, t[b]) for b in t))
print("%-6s %-10s %-14s %-16s %-14s %-12s %-10s" % ("GPUs", "B / rank", "compute ms", "all-reduce ms", "exposed ms", "step ms", "speed-up"))
print("%-6d %-10d %-14.3f %-16s %-14s %-12.3f %-10s" % (1, 520, t[520], "-", "-", t[520], "1.00"))
for n in (2, 4, 8):
    b = 520 // n
    ar = lambda nbytes: 2.0 * (n - 1) / n * nbytes / BUS[n] + LAT
    hidden_window = 0.24e-3 * b / 520.0 + 0.05e-3            # the layer-0 weight-gradient launch at this batch (+ its fixed part)
    exposed = max(0.0, ar(HEAD) - hidden_window) + ar(TAIL)
    step = t[b] * 1e-3 + exposed
    print("%-6d %-10d %-14.3f %-16.3f %-14.3f %-12.3f %-10.2f" % (n, b, t[b], 1e3 * (ar(HEAD) + ar(TAIL)), 1e3 * exposed, 1e3 * step,
                                                              t[520] * 1e-3 / step))
print("weak scaling (520 utterances per GPU, the default of bench.py): compute stays %.3f ms per step; the same exposed all-reduce" % t[520])
for n in (2, 4, 8):
    ar = lambda nbytes: 2.0 * (n - 1) / n * nbytes / BUS[n] + LAT
    exposed = max(0.0, ar(HEAD) - 0.29e-3) + ar(TAIL)
    print("  %d GPUs: %.3f ms per step -> %.2f x the single-GPU throughput" % (n, t[520] + 1e3 * exposed, n * t[520] / (t[520] + 1e3 * exposed)))
