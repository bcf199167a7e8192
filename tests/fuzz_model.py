"""Randomised graphs through the C ABI in f32 mode against the NumPy oracle: 1-4 streams, each with 0-4 encoder layers of
random width and activation, delta layer on / off, LSTM or summed BLSTM per stream, peepholes on / off, every fusion, no
/ forward / bidirectional aggregation LSTM, per-frame or last-timestep head, optional dropout (shared hash masks), ragged
masks, B 1..40, T 1..12.  Forward 2e-5, loss 1e-5, every gradient 1e-4 of the largest gradient tensor's scale; with
`bf16` (the production arithmetic) forward / loss 3e-2 and the whole gradient within cos >= 0.98, norm +-10 %; with
`bf16x3` (fp32-grade GEMMs as three bf16 products; ADN_X3_MIN_WORK=0 sends even these tiny shapes through the split path)
forward 5e-5, loss 2e-5, gradients 5e-4 (a product carries 2^-17 instead of fp32's 2^-24; these graphs use weights of
std 0.3 and widths <= 20, the harshest cancellation this mode meets: observed <= 3e-4, typically 4e-5).

    python tests/fuzz_model.py [n_cases] [seed] [bf16 | bf16x3]      (on an MI355X)"""
import os
import sys

if len(sys.argv) > 3 and sys.argv[3] == "bf16x3":
    os.environ["ADN_X3_MIN_WORK"] = "0"              # (read once, at the library's first GEMM)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ip_avsr_amd.model import AdeNetModel
from oracle import adenet_oracle as O

ACTS = ["rectify", "sigmoid", "tanh", "linear", "leaky_rectify"]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
X3 = len(sys.argv) > 3 and sys.argv[3] == "bf16x3"
BF16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"      # production arithmetic: bf16-MFMA GEMMs and LSTM kernels, looser bounds
bad = 0
for it in range(n_cases):
    S = int(rng.integers(1, 5))
    streams = []
    for k in range(S):
        n_enc = int(rng.integers(0, 5))
        shapes = [int(rng.integers(3, 18)) for _ in range(n_enc)]
        bidir = bool(rng.integers(0, 2)) and S == 1
        sfx = "_s%d" % (k + 1)
        streams.append(dict(input_dim=int(rng.integers(3, 20)), enc_names=[n + sfx for n in O.ENC_NAMES[:n_enc]],
                            enc_shapes=shapes, enc_acts=[str(rng.choice(ACTS)) for _ in range(n_enc)],
                            delta=bool(rng.integers(0, 2)),
                            lstm_names=(["f_lstm" + sfx, "b_lstm" + sfx] if bidir else ["lstm" + sfx]),
                            peepholes=bool(rng.integers(0, 2)), dropout=float(rng.choice([0.0, 0.0, 0.3]))))
    fusion = "none" if S == 1 else str(rng.choice(["sum", "adasum", "concat"]))
    agg = [[], ["lstm_agg"], ["f_lstm_agg", "b_lstm_agg"]][int(rng.integers(0, 3)) if S == 1 else int(rng.integers(1, 3))]
    head = str(rng.choice(["frames", "last"]))
    spec = dict(streams=streams, fusion=fusion, fuse_name={"adasum": "adasum1", "sum": "sum1", "concat": "concat", "none": ""}[fusion],
                agg_names=agg, agg_peepholes=bool(rng.integers(0, 2)), agg_dropout=float(rng.choice([0.0, 0.4])) if agg else 0.0,
                lstm_size=int(rng.integers(2, 12)), classes=int(rng.integers(2, 7)), softmax_name="softmax", head=head,
                loss="cross_entropy" if head == "last" else "temporal")
    B, T, theta = int(rng.integers(1, 41)), int(rng.integers(1, 13)), int(rng.integers(1, 5))
    p = O.init_params(spec, rng, np.float32, enc_std=0.3, perturb=0.03 if BF16 else 0.1)
    lens = rng.integers(1, T + 1, size=B); lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    inputs = [(rng.normal(size=(B, T, s["input_dim"])) * mask[..., None]).astype(np.float32) for s in streams]
    y = np.repeat(rng.integers(0, spec["classes"], size=(B, 1)), T, axis=1).astype(np.int32)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    in64 = [x.astype(np.float64) for x in inputs]
    desc = "S=%d enc=%s delta=%s bidir=%s fusion=%s agg=%d head=%s H=%d C=%d B=%d T=%d theta=%d drop=%s/%.1f" % (
        S, [len(s["enc_shapes"]) for s in streams], [int(s["delta"]) for s in streams], [len(s["lstm_names"]) for s in streams],
        fusion, len(agg), head, spec["lstm_size"], spec["classes"], B, T, theta, [s["dropout"] for s in streams], spec["agg_dropout"])
    try:
        m = AdeNetModel(dict(spec, precision="bf16") if BF16 else dict(spec, precision="bf16x3") if X3 else spec)
        m.set_params_dict(p)
        probs = m.predict(inputs, mask, theta)
        ref = O.forward(spec, p64, in64, mask, theta)
        valid = mask[..., None].astype(bool) if head == "frames" else np.ones((B, 1), bool)
        e_fwd = np.abs((probs - ref) * valid).max()
        dr = dict(seed=1234 + it, counter=it)
        l_ref, g_ref, _ = O.loss_and_grads(spec, p64, in64, y, mask, theta, dropout=dr)
        m.set_dropout_state(dr["seed"], dr["counter"])
        l = m.compute_grads(inputs, y, mask, theta)
        g = m.get_grads_dict()
        gscale = max(np.abs(v).max() for v in g_ref.values())
        e_g = max(np.abs(g[k] - g_ref[k]).max() / max(np.abs(g_ref[k]).max(), 1e-3 * gscale) for k in O.param_names(spec))
        e_l = abs(l - l_ref) / abs(l_ref)
        if BF16:                                              # direction and size of the whole gradient instead of per-tensor maxima
            a = np.concatenate([g[k].ravel() for k in O.param_names(spec)]).astype(np.float64)
            b = np.concatenate([g_ref[k].ravel() for k in O.param_names(spec)])
            cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300)
            e_g = 1.0 - cos
            ok = e_fwd <= 3e-2 and e_l <= 3e-2 and cos >= 0.98 and abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) <= 0.1
        elif X3:
            ok = e_fwd <= 5e-5 and e_l <= 2e-5 and e_g <= 5e-4
        else:
            ok = e_fwd <= 2e-5 and e_l <= 1e-5 and e_g <= 1e-4
        m.close()
    except Exception as ex:                                   # a configuration the library rejects must be reported
        ok, e_fwd, e_l, e_g = False, -1, -1, -1
        desc += "  EXCEPTION " + repr(ex)[:200]
    bad += not ok
    print("%3d %s  fwd %.1e loss %.1e grad %.1e  %s" % (it, "ok " if ok else "BAD", e_fwd, e_l, e_g, desc))
print("bad:", bad)
