"""Randomised shapes (H 5..512, B 1..260, T 1..19, every fusion, peepholes on / off) through the resident-weight LSTM kernels
against the one-workgroup kernels of the same arithmetic: forward identical, gradients within the 19-bit exchange noise,
no exchange time-out.  (Two predictions of one batch must agree bit for bit: the forward GEMMs never split K.)      python tests/fuzz_lstm.py   (on an MI355X)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from oracle import adenet_oracle as O
from test_gpu_parity import make_case
from ip_avsr_amd.model import AdeNetModel
rng = np.random.default_rng(2024)
bad = 0
t0 = time.time()
for it in range(40):
    H = int(rng.choice([5, 37, 64, 100, 250, 256, 257, 300, 500, 512]))
    B = int(rng.choice([1, 2, 31, 32, 33, 63, 65, 100, 260]))
    T = int(rng.integers(1, 20))
    fusion = str(rng.choice(["sum", "concat", "adasum"]))
    peep = bool(rng.integers(0, 2))
    spec = dict(O.spec_nstream([12, 9], enc_shapes=(14, 6), enc_acts=("rectify", "linear"), lstm_size=H, classes=5,
                               fusion=fusion, peepholes=peep), precision="bf16")
    p, inputs, y, mask = make_case(spec, B, T, seed=it, perturb=0.02)
    res = {}
    for mode in ("cluster", "single"):
        os.environ.pop("ADN_LSTM_NO_CLUSTER", None); os.environ.pop("ADN_LSTM_WIDE_PERSISTENT", None)
        if mode == "single":
            os.environ["ADN_LSTM_NO_CLUSTER"] = "1"; os.environ["ADN_LSTM_WIDE_PERSISTENT"] = "1"
        m = AdeNetModel(spec); m.set_params_dict(p)
        pr = m.predict(inputs, mask, 2); l = m.compute_grads(inputs, y, mask, 2); g = m.get_grads_dict()
        pr2 = m.predict(inputs, mask, 2)
        if not (pr == pr2).all():
            bad += 1
            print('   NON-REPEATABLE predict: it=%d mode=%s H=%d B=%d T=%d %s peep=%d  max diff %.3e (valid %.3e)' % (it, mode, H, B, T, fusion, peep, np.abs(pr-pr2).max(), np.abs((pr-pr2)*mask[...,None]).max()))
        res[mode] = (pr, l, g); m.close()
    valid = mask[..., None].astype(bool)
    dp = np.abs((res["cluster"][0] - res["single"][0]) * valid).max()
    dg = max(np.abs(res["cluster"][2][k] - res["single"][2][k]).max() / max(np.abs(res["single"][2][k]).max(), 1e-6) for k in res["single"][2])
    ok = dp == 0 and dg <= 5e-3 and np.isfinite(res["cluster"][1])
    bad += not ok
    print("%2d H=%3d B=%3d T=%2d %-6s peep=%d  dprobs %.1e dgrads %.1e %s" % (it, H, B, T, fusion, peep, dp, dg, "ok" if ok else "BAD"))
print("bad:", bad, "time %.1f s" % (time.time() - t0))
