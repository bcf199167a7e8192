"""Per-GPU step time of the data parallel code path on ONE GPU, no collective: (a) the plain step, (b) gradient buckets +
events registered in the layer-major order (default), (c) the same with Adam applied bucket by bucket, (d) the stream-major
order (ADN_DP_STREAM_MAJOR=1).  bf16 mode, bfloat16-resident batch like the headline.
python profiles/scripts/dp_order_bench.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RUN = r'''
import os, sys, torch
sys.path.insert(0, %r)
import bench
from ip_avsr_amd.model import AdeNetModel
m = AdeNetModel(bench.build_spec()); m.set_precision("bf16"); bench.synthetic_params(m)
xs, y, m_d, _ = bench.synthetic_batch(torch, 0, bench.B_PER_GPU, torch.device("cuda", 0))
xs = [x.to(torch.bfloat16) for x in xs]
evs = []
buckets = m.grad_buckets()
if os.environ.get("BUCKETS"):
    for _ in buckets:
        e = torch.cuda.Event(); e.record(); evs.append(e)
    m.set_bucket_events([e.cuda_event for e in evs])
ranged = bool(os.environ.get("RANGED"))
def step():
    m.compute_grads(xs, y, m_d, bench.THETA, want_loss=False)
    if ranged:
        m.adam_begin(2e-3)
        for b, e in buckets: m.adam_range(b, e)
        m.adam_end()
    else:
        m.apply_adam(2e-3)
for _ in range(30): step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(40): step()
b.record(); torch.cuda.synchronize()
print("%%d buckets, %%.3f ms per step" %% (len(buckets), a.elapsed_time(b) / 40))
''' % ROOT
for name, env in (("plain step (no bucket events)", {}),
                  ("layer-major (default), events registered", {"BUCKETS": "1"}),
                  ("layer-major, events + Adam per bucket", {"BUCKETS": "1", "RANGED": "1"}),
                  ("stream-major (ADN_DP_STREAM_MAJOR=1), events registered", {"BUCKETS": "1", "ADN_DP_STREAM_MAJOR": "1"})):
    out = subprocess.run([sys.executable, "-c", RUN], env=dict(os.environ, **env), capture_output=True, text=True)
    print(name + ":", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
