"""Per-GPU step time of the data parallel code path (gradient buckets + events registered, no collective: one GPU) in the
layer-major order (default) and the stream-major one (ADN_DP_STREAM_MAJOR=1).  python profiles/scripts/dp_order_bench.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RUN = r'''
import sys, torch
sys.path.insert(0, %r)
import bench
from ip_avsr_amd.model import AdeNetModel
m = AdeNetModel(bench.build_spec()); m.set_precision("bf16"); bench.synthetic_params(m)
xs, y, m_d, _ = bench.synthetic_batch(torch, 0, bench.B_PER_GPU, torch.device("cuda", 0))
evs = []
for _ in m.grad_buckets():
    e = torch.cuda.Event(); e.record(); evs.append(e)
m.set_bucket_events([e.cuda_event for e in evs])
def step():
    m.compute_grads(xs, y, m_d, bench.THETA, want_loss=False); m.apply_adam(2e-3)
for _ in range(30): step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): step()
b.record(); torch.cuda.synchronize()
print("%%.3f ms per step" %% (a.elapsed_time(b) / 20))
''' % ROOT
for name, env in (("layer-major (default)", {}), ("stream-major (ADN_DP_STREAM_MAJOR=1)", {"ADN_DP_STREAM_MAJOR": "1"})):
    out = subprocess.run([sys.executable, "-c", RUN], env=dict(os.environ, **env), capture_output=True, text=True)
    print(name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
