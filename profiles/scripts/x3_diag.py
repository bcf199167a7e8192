import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from ip_avsr_amd.model import AdeNetModel
m = AdeNetModel(bench.build_spec())
bench.synthetic_params(m)
xs, y, m_d, mask = bench.synthetic_batch(torch, 0, bench.B_PER_GPU, torch.device("cuda", 0))
for _ in range(2):
    m.train_step(xs, y, m_d, bench.THETA, 2e-3, want_loss=False)
def grads(prec):
    m.set_precision(prec)
    m.compute_grads(xs, y, m_d, bench.THETA)
    return m.get_grads_dict()
a, a2, b = grads("f32"), grads("f32"), grads("bf16x3")
rel = lambda u, v: np.linalg.norm(u.astype(np.float64).ravel() - v.astype(np.float64).ravel()) / max(np.linalg.norm(u.astype(np.float64).ravel()), 1e-300)
for k in a:
    print("%-34s f32-vs-f32 %.2e   f32-vs-x3 %.2e   |g| %.2e" % (k, rel(a[k], a2[k]), rel(a[k], b[k]), np.abs(a[k]).max()))
