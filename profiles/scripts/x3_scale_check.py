import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
os.environ.pop("ADN_LSTM_NO_X3_CLUSTER", None)
from tests.test_gpu_bf16x3 import _small_x3_model, ragged_mask
from oracle import adenet_oracle as O
for H, B, T in [(512, 96, 40), (512, 520, 12), (250, 520, 40), (250, 96, 40), (512, 96, 12)]:
    spec, p, m, rng = _small_x3_model(H, True, 100 * H + B + T)
    mask = ragged_mask(rng, B, T)
    xs = [(rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32) for d in (60, 44)]
    y = np.repeat((np.arange(B) % 26)[:, None], T, axis=1).astype(np.int32)
    res = {}
    for mode in ("cluster", "steps"):
        if mode == "steps": os.environ["ADN_LSTM_NO_X3_CLUSTER"] = "1"
        else: os.environ.pop("ADN_LSTM_NO_X3_CLUSTER", None)
        res[mode] = (m.predict(xs, mask, 9), m.compute_grads(xs, y, mask, 9), m.get_grads_dict())
    os.environ.pop("ADN_LSTM_NO_X3_CLUSTER", None)
    dp = np.abs(res["cluster"][0] - res["steps"][0]).max()
    gs = max(np.abs(v).max() for v in res["steps"][2].values())
    worst = max(np.abs(res["cluster"][2][k] - g).max() / max(np.abs(g).max(), 1e-3 * gs) for k, g in res["steps"][2].items())
    line = "H=%d B=%d T=%d: cluster vs steps |dp| %.1e grads %.1e" % (H, B, T, dp, worst)
    if B * T <= 4000:
        p64 = {k: v.astype(np.float64) for k, v in p.items()}
        ref = O.forward(spec, p64, [x.astype(np.float64) for x in xs], mask, 9)
        line += " | vs fp64 oracle: cluster %.1e, steps %.1e" % (np.abs(res["cluster"][0] - ref).max(), np.abs(res["steps"][0] - ref).max())
    print(line, flush=True)
    m.close()
