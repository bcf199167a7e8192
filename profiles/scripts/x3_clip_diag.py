import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import adenet_oracle as O
import test_gpu_bf16x3 as Tm
spec, p, m, rng = Tm._small_x3_model(48, True, 21)
B, T, theta = 37, 8, 2
mask = Tm.ragged_mask(rng, B, T)
xs = [(rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32) for d in (60, 44)]
y = np.repeat((np.arange(B) % 26)[:, None], T, axis=1).astype(np.int32)
q = {k: v.copy() for k, v in p.items()}
q["softmax.W"] = (rng.normal(size=q["softmax.W"].shape) * 4000).astype(np.float32)
q64 = {k: v.astype(np.float64) for k, v in q.items()}
x64 = [x.astype(np.float64) for x in xs]
_, g_ref, _ = O.loss_and_grads(spec, q64, x64, y, mask, theta)
q32 = {k: v.astype(np.float32) for k, v in q.items()}
_, g_ref32, _ = O.loss_and_grads(spec, q32, xs, y, mask, theta)
keys = ("f_lstm_agg.b_ingate", "b_lstm_agg.b_outgate", "lstm_s1.W_hid_to_cell", "lstm_s2.W_cell_to_outgate", "softmax.W", "fc1_s1.W")
print("oracle fp32 vs fp64:", {k: float(np.abs(g_ref32[k] - g_ref[k]).max() / np.abs(g_ref[k]).max()) for k in keys})
for prec, env in (("f32", None), ("bf16x3", "ADN_LSTM_NO_X3_CLUSTER"), ("bf16x3", None)):
    if env: os.environ[env] = "1"
    m.set_precision(prec)
    m.set_params_dict(q)
    m.compute_grads(xs, y, mask, theta)
    g = m.get_grads_dict()
    print(prec, env, {k: float(np.abs(g[k] - g_ref[k]).max() / np.abs(g_ref[k]).max()) for k in keys})
    if env: os.environ.pop(env)
