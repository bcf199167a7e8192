"""Which utterances / frames differ between the weight-stationary and the one-workgroup LSTM forward at the bench geometry?  (debug aid)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RUN = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import bench
from ip_avsr_amd.model import AdeNetModel
torch.cuda.set_device(0)
m = AdeNetModel(bench.build_spec()); m.set_precision("bf16"); bench.synthetic_params(m)
xs, y, m_d, mask = bench.synthetic_batch(torch, 0, 520, torch.device("cuda", 0))
if len(sys.argv) > 2:
    saved = np.load(sys.argv[2])
    for p in m.params: p.set_value(saved["p_" + p.name])
else:
    for _ in range(3): m.train_step(xs, y, m_d, bench.THETA, 2e-3, want_loss=False)
params = {"p_" + p.name: p.get_value() for p in m.params}
probs = [m.predict(xs, m_d, bench.THETA) for _ in range(3)]
np.savez(sys.argv[1], probs=np.stack(probs), mask=mask, **params)
''' % ROOT
a = "/tmp/ld_default.npz"; b = "/tmp/ld_nocluster.npz"
subprocess.run([sys.executable, "-c", RUN, a], check=True)
subprocess.run([sys.executable, "-c", RUN, b, a], check=True, env=dict(os.environ, ADN_LSTM_NO_CLUSTER="1"))
A, B = np.load(a), np.load(b)
mask = A["mask"].astype(bool)
print("repeat-to-repeat (cluster):", [int((A["probs"][0] != A["probs"][k]).sum()) for k in (1, 2)], " (one-workgroup):", [int((B["probs"][0] != B["probs"][k]).sum()) for k in (1, 2)])
d = (A["probs"][0] != B["probs"][0]).any(-1) & mask
rows = np.nonzero(d.any(1))[0]
lens = mask.sum(1)
print("utterances that differ:", len(rows), "of", len(mask), "; lengths of those:", sorted(set(lens[rows].tolist())), "; all lengths:", sorted(set(lens.tolist()))[:12], "...")
print("rows:", rows[:60].tolist())
print("first differing frame per row (first 20):", [(int(r), int(np.nonzero(d[r])[0][0]), int(lens[r])) for r in rows[:20]])
