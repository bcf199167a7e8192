"""CPU oracle for the AdeNet / DeltaNet hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference's training step.  It is the
checker for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``ip_avsr_amd``) never imports anything under ``oracle/``.

PARITY STATUS: *parity unpinned at the Theano/Lasagne boundary*.  The reference
delegates all hot-path arithmetic to Theano + Lasagne (unpinned ``master.zip``,
README.md:30-34), which are absent from /root/reference and cannot be installed
here; no reference test pins a floating-point output of this path.
What pins this oracle instead:
  * finite-difference checks of every backward (tests/test_oracle.py, float64);
  * ``torch.nn.LSTM`` on CPU as an independent second opinion for the recurrence
    (same i,f,g,o gate order), incl. gradients;
  * the closed-form delta operator against a literal transcription of the
    reference's three nested scans (utils/signal.py:7-80) and the derived
    known answers of SURVEY.md §8c;
  * the host-side NumPy functions are pinned separately against golden vectors
    produced by importing the reference (tests/golden/make_golden.py).

Each function cites the reference file:line it restates.  Lasagne semantics
(``[upstream]``) follow SURVEY.md Appendix A.
"""
from __future__ import annotations

import numpy as np

GRAD_CLIP = 5.0  # grad_clipping=5. in every LSTMLayer, e.g. modelzoo/adenet_v2.py:53
GATES = ("ingate", "forgetgate", "cell", "outgate")  # stacking order [upstream] App. A-3


# --------------------------------------------------------------------------- #
# nonlinearities: custom/nonlinearities.py:4-16 -> lasagne.nonlinearities
# --------------------------------------------------------------------------- #
def act_fwd(name, z):
    if name in ("linear", "identity"):
        return z
    if name == "rectify":
        return np.maximum(z, 0)
    if name == "sigmoid":
        return 1.0 / (1.0 + np.exp(-z))
    if name == "tanh":
        return np.tanh(z)
    if name == "leaky_rectify":
        return np.where(z > 0, z, z * z.dtype.type(0.01))
    if name == "very_leaky_rectify":
        return np.where(z > 0, z, z * z.dtype.type(1.0 / 3.0))
    raise ValueError("unsupported nonlinearity %r" % (name,))


def act_bwd(name, y, dy, kink=None, at_zero=0.0):
    """dL/dz from the *output* y (every supported act is invertible enough).

    The rectifier at a pre-activation of EXACTLY zero.  Lasagne's ``rectify`` is Theano's ``0.5 * (x + abs(x))`` [upstream], and
    Theano differentiates ``abs`` as ``sgn(x)`` with ``sgn(0) = 0``: the reference's relu'(0) is 0.5.  That is NOT a measure-zero
    case for this model: the minibatch generators pad with zero frames (utils/datagen.py:129-142), so with zero encoder biases
    (SURVEY.md 8d's synthetic parameters) every padding row has pre-activation exactly 0 in every rectifier layer, and the delta
    layer leaks gradient into those rows (App. E-2) -- the reference sends half of it on into the biases below.  With DBN-pretrained
    (non-zero) biases nothing sits on the kink.  ``at_zero`` selects the convention: 0.0 (this build's default: the mask is
    ``y > 0``) or 0.5 (Theano's); ``kink`` is the boolean mask of the exactly-zero pre-activations kept by ``forward``
    (spec["relu_grad_at_zero"]; the HIP path: adn_set_relu_grad_at_zero)."""
    if name in ("linear", "identity"):
        return dy
    if name == "rectify":
        g = (y > 0).astype(dy.dtype)
        if at_zero and kink is not None:
            g = g + dy.dtype.type(at_zero) * kink
        return dy * g
    if name == "sigmoid":
        return dy * y * (1 - y)
    if name == "tanh":
        return dy * (1 - y * y)
    if name == "leaky_rectify":
        return np.where(y > 0, dy, dy * y.dtype.type(0.01))
    if name == "very_leaky_rectify":
        return np.where(y > 0, dy, dy * y.dtype.type(1.0 / 3.0))
    raise ValueError(name)


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


# --------------------------------------------------------------------------- #
# Delta layer: custom/layers.py:105-121 -> utils/signal.py:59-80,26-39,7-23
# --------------------------------------------------------------------------- #
def delta_literal(A, theta):
    """Literal transcription of ``delta_coeff`` (utils/signal.py:42-56) for one
    (T,F) float32 sequence, including Theano's int32*float32 -> float64 upcast of
    every term and the float32 cast of the accumulator after each theta
    (utils/signal.py:19-21).  O(T*Theta*F); small cases only."""
    A = np.asarray(A, dtype=np.float32)
    T, F = A.shape
    X = A.T
    Y = np.concatenate([np.repeat(X[:, :1], theta, axis=1), X,
                        np.repeat(X[:, -1:], theta, axis=1)], axis=1)
    out = np.zeros((T, F), np.float32)
    for t in range(T):
        acc = np.zeros((F,), np.float32)
        for th in range(1, theta + 1):
            d = th * (Y[:, theta + t + th].astype(np.float64)
                      - Y[:, theta + t - th].astype(np.float64)) / (2 * th * th)
            acc = (acc.astype(np.float64) + d).astype(np.float32)
        out[t] = acc
    return out


def append_delta_literal(A, theta):
    """``append_delta_coeff`` (utils/signal.py:59-80): [A | delta(A) | delta(delta(A))]."""
    d1 = delta_literal(A, theta)
    d2 = delta_literal(d1, theta)
    return np.concatenate([np.asarray(A, np.float32), d1, d2], axis=1)


def delta_op(x, theta):
    """Closed form of the delta operator along axis 1 of (B,T,F), any float dtype:
    ``d[t] = sum_{k=1..theta} (x[clamp(t+k)] - x[clamp(t-k)]) / (2k)`` with indices
    clamped to the *tensor* ends [0, T-1] (mask-blind)."""
    B, T, F = x.shape
    idx = np.arange(T)
    out = np.zeros_like(x)
    for k in range(1, theta + 1):
        hi = np.minimum(idx + k, T - 1)
        lo = np.maximum(idx - k, 0)
        out = out + (x[:, hi] - x[:, lo]) * x.dtype.type(1.0 / (2 * k))
    return out


def delta_op_T(g, theta):
    """Adjoint of ``delta_op`` (scatter form)."""
    B, T, F = g.shape
    idx = np.arange(T)
    out = np.zeros_like(g)
    for k in range(1, theta + 1):
        hi = np.minimum(idx + k, T - 1)
        lo = np.maximum(idx - k, 0)
        w = g * g.dtype.type(1.0 / (2 * k))
        np.add.at(out, (slice(None), hi), w)
        np.add.at(out, (slice(None), lo), -w)
    return out


def delta_append(x, theta):
    d1 = delta_op(x, theta)
    d2 = delta_op(d1, theta)
    return np.concatenate([x, d1, d2], axis=2)


def delta_append_bwd(dout, theta):
    F = dout.shape[2] // 3
    g0, g1, g2 = dout[..., :F], dout[..., F:2 * F], dout[..., 2 * F:]
    return g0 + delta_op_T(g1 + delta_op_T(g2, theta), theta)


# --------------------------------------------------------------------------- #
# LSTMLayer [upstream] as configured by custom/layers.py:10-25,55-80 and inline
# in modelzoo/adenet_v2.py:45-63 -- semantics SURVEY App. A-3
# --------------------------------------------------------------------------- #
def _stack(p, prefix, kind):
    return np.concatenate([p["%s.%s_to_%s" % (prefix, kind, g)] for g in GATES], axis=1)


def lstm_fwd(x, mask, p, name, backwards=False, peepholes=False):
    """x (B,T,F), mask (B,T) {0,1}.  Returns h_seq (B,T,H) and a cache."""
    dt = x.dtype
    B, T, F = x.shape
    W_in = _stack(p, name, "W_in")
    W_hid = _stack(p, name, "W_hid")
    b = np.concatenate([p["%s.b_%s" % (name, g)] for g in GATES])
    H = W_hid.shape[0]
    xproj = (x.reshape(B * T, F) @ W_in + b).reshape(B, T, 4 * H)  # precompute_input=True
    ones = np.ones((B, 1), dt)
    h_prev = ones @ p[name + ".hid_init"]      # learn_init=True: (1,H) broadcast via dot
    c_prev = ones @ p[name + ".cell_init"]
    if peepholes:
        w_ci, w_cf, w_co = (p["%s.W_cell_to_%s" % (name, g)] for g in ("ingate", "forgetgate", "outgate"))
    hs = np.zeros((B, T, H), dt)
    cache = dict(x=x, mask=mask, steps=[], W_in=W_in, W_hid=W_hid, H=H,
                 backwards=backwards, peepholes=peepholes, name=name)
    order = range(T - 1, -1, -1) if backwards else range(T)
    for t in order:
        gates = xproj[:, t] + h_prev @ W_hid
        a_i, a_f, a_g, a_o = (gates[:, k * H:(k + 1) * H] for k in range(4))
        if peepholes:
            a_i = a_i + c_prev * w_ci
            a_f = a_f + c_prev * w_cf
        i, f, g = sigmoid(a_i), sigmoid(a_f), np.tanh(a_g)
        c_new = f * c_prev + i * g
        if peepholes:
            a_o = a_o + c_new * w_co
        o = sigmoid(a_o)
        h_new = o * np.tanh(c_new)
        m = mask[:, t].astype(bool)[:, None]
        c = np.where(m, c_new, c_prev)
        h = np.where(m, h_new, h_prev)
        cache["steps"].append((t, i, f, g, o, c_new, c_prev, h_prev))
        hs[:, t] = h
        h_prev, c_prev = h, c
    return hs, cache


def lstm_bwd(dhs, cache, p, grads):
    """BPTT with the gate-pre-activation gradient clipped to +-5 each step
    (theano.gradient.grad_clip on ``gates`` before the peephole adds) [upstream]."""
    name, H = cache["name"], cache["H"]
    x, mask, W_in, W_hid = cache["x"], cache["mask"], cache["W_in"], cache["W_hid"]
    peep = cache["peepholes"]
    dt = x.dtype
    B, T, F = x.shape
    if peep:
        w_ci, w_cf, w_co = (p["%s.W_cell_to_%s" % (name, g)] for g in ("ingate", "forgetgate", "outgate"))
        dw_ci = np.zeros(H, dt); dw_cf = np.zeros(H, dt); dw_co = np.zeros(H, dt)
    dxproj = np.zeros((B, T, 4 * H), dt)
    dW_hid = np.zeros_like(W_hid)
    dh = np.zeros((B, H), dt)
    dc = np.zeros((B, H), dt)
    for (t, i, f, g, o, c_new, c_prev, h_prev) in reversed(cache["steps"]):
        dh = dh + dhs[:, t]
        m = mask[:, t].astype(dt)[:, None]
        dh_c, dh_p = dh * m, dh * (1 - m)
        dc_c, dc_p = dc * m, dc * (1 - m)
        tc = np.tanh(c_new)
        da_o = dh_c * tc * o * (1 - o)
        dcn = dc_c + dh_c * o * (1 - tc * tc)
        if peep:
            dcn = dcn + da_o * w_co
            dw_co += (da_o * c_new).sum(0)
        da_i = dcn * g * i * (1 - i)
        da_f = dcn * c_prev * f * (1 - f)
        da_g = dcn * i * (1 - g * g)
        dc_p = dc_p + dcn * f
        if peep:
            dc_p = dc_p + da_i * w_ci + da_f * w_cf
            dw_ci += (da_i * c_prev).sum(0)
            dw_cf += (da_f * c_prev).sum(0)
        dgates = np.clip(np.concatenate([da_i, da_f, da_g, da_o], axis=1), -GRAD_CLIP, GRAD_CLIP)
        dxproj[:, t] = dgates
        dW_hid += h_prev.T @ dgates
        dh = dh_p + dgates @ W_hid.T
        dc = dc_p
    grads[name + ".hid_init"] = dh.sum(0, keepdims=True)
    grads[name + ".cell_init"] = dc.sum(0, keepdims=True)
    dxp = dxproj.reshape(B * T, 4 * H)
    dW_in = x.reshape(B * T, F).T @ dxp
    db = dxp.sum(0)
    for k, gname in enumerate(GATES):
        grads["%s.W_in_to_%s" % (name, gname)] = dW_in[:, k * H:(k + 1) * H]
        grads["%s.W_hid_to_%s" % (name, gname)] = dW_hid[:, k * H:(k + 1) * H]
        grads["%s.b_%s" % (name, gname)] = db[k * H:(k + 1) * H]
    if peep:
        grads[name + ".W_cell_to_ingate"] = dw_ci
        grads[name + ".W_cell_to_forgetgate"] = dw_cf
        grads[name + ".W_cell_to_outgate"] = dw_co
    return (dxp @ W_in.T).reshape(B, T, F)


# --------------------------------------------------------------------------- #
# loss: custom/objectives.py:4-39 (double softmax)
# --------------------------------------------------------------------------- #
def softmax_rows(z):
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def temporal_softmax_loss(x, y, mask, total_frames=None):
    """x (B,T,C) are *already probabilities*; they are soft-maxed again.
    ``total_frames`` overrides the normaliser (data parallel: the global batch's valid frames)."""
    N = x.shape[0] * x.shape[1]
    xf = x.reshape(N, -1)
    q = softmax_rows(xf)
    mf = mask.reshape(N).astype(x.dtype)
    total = mf.sum() if total_frames is None else x.dtype.type(total_frames)
    return -(mf * np.log(q[np.arange(N), y.reshape(N)])).sum() / total


def temporal_softmax_loss_bwd(x, y, mask, total_frames=None):
    N = x.shape[0] * x.shape[1]
    q = softmax_rows(x.reshape(N, -1))
    mf = mask.reshape(N).astype(x.dtype)
    q[np.arange(N), y.reshape(N)] -= 1
    total = mf.sum() if total_frames is None else x.dtype.type(total_frames)
    return (q * (mf / total)[:, None]).reshape(x.shape)


# --------------------------------------------------------------------------- #
# model spec + parameter order (Lasagne get_all_params order)
# --------------------------------------------------------------------------- #
def lstm_param_names(name, peepholes):
    out = []
    for g in GATES:
        out += ["%s.W_in_to_%s" % (name, g), "%s.W_hid_to_%s" % (name, g), "%s.b_%s" % (name, g)]
    if peepholes:
        out += ["%s.W_cell_to_%s" % (name, g) for g in ("ingate", "forgetgate", "outgate")]
    out += [name + ".cell_init", name + ".hid_init"]
    return out


def param_names(spec):
    """spec: dict(streams=[dict(input_dim, enc_names, enc_shapes, enc_acts, delta,
    lstm_names=[...1 or 2 (f,b)], peepholes)], fusion, fuse_name, agg_names=[f,b] or [],
    agg_peepholes, lstm_size, classes, softmax_name)."""
    names = []
    for s in spec["streams"]:
        for n in s["enc_names"]:
            names += [n + ".W", n + ".b"]
        if s.get("batchnorm"):                       # lasagne BatchNormLayer registers beta, gamma, mean, inv_std
            names += ["%s.%s" % (s["batchnorm"], k) for k in BN_PARAMS]
        for ln in s["lstm_names"]:
            names += lstm_param_names(ln, s["peepholes"])
    if spec["fusion"] == "adasum":
        names += ["%s.adacoeff%d" % (spec["fuse_name"], k) for k in range(len(spec["streams"]))]
    for ln in spec["agg_names"]:
        names += lstm_param_names(ln, spec["agg_peepholes"])
    names += [spec["softmax_name"] + ".W", spec["softmax_name"] + ".b"]
    return names


BN_PARAMS = ("beta", "gamma", "mean", "inv_std")
BN_EPS, BN_ALPHA = 1e-4, 0.1                             # lasagne.layers.BatchNormLayer defaults


def lstm_in_dim(spec, s):
    """width of what the stream's LSTM reads: [x | dx | ddx] of the encoder output, then the stream's auxiliary input
    (ConcatLayer([l_delta, l_dct], axis=2), modelzoo/adenet_v1.py:87)"""
    d = s["enc_shapes"][-1] if s["enc_shapes"] else s["input_dim"]
    return (d * 3 if s["delta"] else d) + int(s.get("aux_dim", 0) or 0)


def param_shapes(spec):
    """``stream_lstm_size`` (optional): units of the STREAM LSTMs when they differ from the aggregation LSTMs'
    ``lstm_size`` (modelzoo/adenet_v1.py:89,95: lstm_size and 2 * lstm_size)."""
    H, shapes = spec["lstm_size"], {}
    Hs = int(spec.get("stream_lstm_size") or H)

    def lstm(name, fin, peep, H=H):
        for g in GATES:
            shapes["%s.W_in_to_%s" % (name, g)] = (fin, H)
            shapes["%s.W_hid_to_%s" % (name, g)] = (H, H)
            shapes["%s.b_%s" % (name, g)] = (H,)
        if peep:
            for g in ("ingate", "forgetgate", "outgate"):
                shapes["%s.W_cell_to_%s" % (name, g)] = (H,)
        shapes[name + ".cell_init"] = (1, H)
        shapes[name + ".hid_init"] = (1, H)

    for s in spec["streams"]:
        d = s["input_dim"]
        for n, u in zip(s["enc_names"], s["enc_shapes"]):
            shapes[n + ".W"] = (d, u)
            shapes[n + ".b"] = (u,)
            d = u
        if s.get("batchnorm"):
            for k in BN_PARAMS:
                shapes["%s.%s" % (s["batchnorm"], k)] = (d,)
        for ln in s["lstm_names"]:
            lstm(ln, lstm_in_dim(spec, s), s["peepholes"], Hs)
    S = len(spec["streams"])
    if spec["fusion"] == "adasum":
        for k in range(S):
            shapes["%s.adacoeff%d" % (spec["fuse_name"], k)] = ()
    fused = Hs * S if spec["fusion"] == "concat" else Hs
    for ln in spec["agg_names"]:
        lstm(ln, fused, spec["agg_peepholes"])
    shapes[spec["softmax_name"] + ".W"] = (H, spec["classes"])
    shapes[spec["softmax_name"] + ".b"] = (spec["classes"],)
    return shapes


def init_params(spec, rng, dtype=np.float32, enc_std=0.01, perturb=0.0):
    """Synthetic parameters per SURVEY §8d: encoder N(0,enc_std) weights, LSTM /
    softmax GlorotUniform, biases 0 (+ optional perturbation so that no parameter
    sits at a degenerate value in gradient checks)."""
    shapes, p = param_shapes(spec), {}
    for n in param_names(spec):
        shp = shapes[n]
        leaf = n.split(".")[-1]
        if leaf == "W" and not n.startswith(spec["softmax_name"] + "."):
            v = rng.normal(0, enc_std, shp)
        elif len(shp) == 2 and shp[0] > 1:
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            v = rng.uniform(-lim, lim, shp)
        elif leaf.startswith("adacoeff"):
            v = np.ones(shp)
        elif leaf.startswith("W_cell_to"):
            v = rng.normal(0, 0.1, shp)      # Gate() default W_cell=Normal(0.1) [upstream] App. A-2
        elif leaf in ("gamma", "inv_std"):
            v = np.ones(shp)                 # BatchNormLayer: gamma = 1, inv_std = 1 (beta = mean = 0)
            if perturb:
                v = v + np.abs(rng.normal(0, perturb, shp)) - rng.normal(0, perturb, shp)
        else:
            v = np.zeros(shp)
        if perturb:
            v = v + rng.normal(0, perturb, shp)
        p[n] = np.asarray(v, dtype)
    return p


# --------------------------------------------------------------------------- #
# whole model: forward / loss / backward / Adam
# --------------------------------------------------------------------------- #
# --------------------------------------------------------------------------- #
# DropoutLayer [upstream lasagne.layers.DropoutLayer, rescale=True]: out = x * mask / (1 - p) with
# mask ~ Bernoulli(1 - p) per element of the (B,T,F) tensor (modelzoo/adenet_v3.py:112,123,134,154).
# Theano's MRG stream cannot be reproduced; the mask is DEFINED here by a counter-based integer hash of
# (seed, call counter, layer id, element index in (B,T,F) C order) so that the HIP path and this oracle draw
# the same mask bit for bit (csrc/elementwise.hip::dropout_keep).
# --------------------------------------------------------------------------- #
DROPOUT_AGG_LAYER = 100


def dropout_uniform(seed, counter, layer, idx):
    """uint32 hash -> uniform in [0,1) with 24 bits (idx: array of element indices)."""
    with np.errstate(over="ignore"):
        k = np.uint32(seed & 0xFFFFFFFF) ^ np.uint32((layer * 0x85EBCA77) & 0xFFFFFFFF) ^ \
            np.uint32((counter * 0xC2B2AE3D) & 0xFFFFFFFF)
        x = np.asarray(idx).astype(np.uint32) * np.uint32(0x9E3779B1) + k
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return (x >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)


def dropout_scale(shape, prob, dropout, layer, dtype):
    """mask / (1 - p) for a (B,T,F) tensor; ones when deterministic (dropout is None) or p == 0."""
    if dropout is None or not prob:
        return np.ones(shape, dtype)
    idx = np.arange(int(np.prod(shape)), dtype=np.uint64).reshape(shape)
    keep = dropout_uniform(dropout["seed"], dropout.get("counter", 0), layer, idx) >= np.float32(prob)
    return keep.astype(dtype) * (dtype(1) / (dtype(1) - dtype(prob)))


def forward(spec, p, inputs, mask, theta, want_cache=False, dropout=None, training=None):
    """inputs: list of (B,T,D_s), followed by the auxiliary inputs (B,T,A_s) of the streams that have one, in stream
    order.  Returns probs (B,T,C) -- (B,C) for the last-timestep head -- [, cache].
    dropout: None = deterministic; dict(seed=, counter=) = stochastic layers active.
    training: BatchNorm layers use batch statistics (get_output(deterministic=False)); default: dropout is not None.
    Graph: modelzoo/adenet_v2.py:30-92, adenet_3stream.py:166-262, adenet_4stream.py:37-157,
    avnet.py:43-112, deltanet_majority_vote.py:31-66 (S=1, no aggregation layer)."""
    B, T = mask.shape
    H = spec["lstm_size"]
    if training is None:
        training = dropout is not None
    cache = dict(streams=[], training=training)
    outs = []
    S_ = len(spec["streams"])
    aux_inputs = list(inputs[S_:])
    for s, x in zip(spec["streams"], inputs[:S_]):
        sc = dict(acts=[x.reshape(B * T, -1)], kinks=[])
        a = sc["acts"][0]
        for n, act in zip(s["enc_names"], s["enc_acts"]):   # modelzoo/pretrained_encoder.py:4-9
            z = a @ p[n + ".W"] + p[n + ".b"]
            a = act_fwd(act, z)
            sc["acts"].append(a)
            sc["kinks"].append((z == 0) if act == "rectify" else None)       # (act_bwd: relu'(0))
        if s.get("batchnorm"):                               # BatchNormLayer on the (B*T, E) encoder output (adenet_v1.py:82)
            bn = s["batchnorm"]
            if training:
                mu = a.mean(0)
                inv = 1.0 / np.sqrt(((a - mu) ** 2).mean(0) + a.dtype.type(BN_EPS))
            else:
                mu, inv = p[bn + ".mean"], p[bn + ".inv_std"]
            sc["bn"] = dict(x=a, mean=mu, inv_std=inv, xhat=(a - mu) * inv)
            a = sc["bn"]["xhat"] * p[bn + ".gamma"] + p[bn + ".beta"]
        feat = a.reshape(B, T, -1)
        sc["enc_out"] = feat
        if s["delta"]:
            feat = delta_append(feat, theta)
        if s.get("aux_dim"):                                 # ConcatLayer([l_delta, l_dct], axis=2) (adenet_v1.py:87)
            feat = np.concatenate([feat, np.asarray(aux_inputs.pop(0), feat.dtype)], axis=2)
        sc["drop"] = dropout_scale(feat.shape, s.get("dropout", 0.0), dropout, len(outs), feat.dtype.type)
        feat = feat * sc["drop"]                              # DropoutLayer ahead of the stream LSTM (adenet_v3.py:112)
        sc["lstm_in"] = feat
        h = None
        sc["lstm"] = []
        for k, ln in enumerate(s["lstm_names"]):          # 1 = LSTM, 2 = BLSTM summed
            hk, lc = lstm_fwd(feat, mask, p, ln, backwards=(k == 1), peepholes=s["peepholes"])
            sc["lstm"].append(lc)
            h = hk if h is None else h + hk
        sc["h"] = h
        outs.append(h)
        cache["streams"].append(sc)
    fusion = spec["fusion"]
    if fusion == "concat":                                   # ConcatLayer(axis=-1)
        fused = np.concatenate(outs, axis=2)
    elif fusion == "sum":
        fused = sum(outs)
    elif fusion == "adasum":                                 # custom/layers.py:178-228
        fused = sum(o * p["%s.adacoeff%d" % (spec["fuse_name"], k)] for k, o in enumerate(outs))
    elif fusion == "none":
        fused = outs[0]
    else:
        raise ValueError(fusion)
    cache["fused_drop"] = dropout_scale(fused.shape, spec.get("agg_dropout", 0.0), dropout, DROPOUT_AGG_LAYER,
                                        fused.dtype.type)
    fused = fused * cache["fused_drop"]                       # dropout_agg (adenet_v3.py:154)
    cache["fused"] = fused
    cache["agg"] = []
    if spec["agg_names"]:
        hsum = None
        for k, ln in enumerate(spec["agg_names"]):          # custom/layers.py:55-80 create_blstm
            hk, lc = lstm_fwd(fused, mask, p, ln, backwards=(k == 1), peepholes=spec["agg_peepholes"])
            cache["agg"].append(lc)
            hsum = hk if hsum is None else hsum + hk
    else:
        hsum = fused
    cache["hsum"] = hsum
    sm = spec["softmax_name"]
    if spec.get("head", "frames") == "last":
        # SliceLayer(l_sum2, -1, 1) + DenseLayer(softmax) (adenet_v3.py:180-186, deltanet.py:48-56): the LAST row of
        # the padded tensor -- held last-valid state of the forward LSTM, first step of the backward one (App. E-3)
        probs = softmax_rows(hsum[:, T - 1, :] @ p[sm + ".W"] + p[sm + ".b"])
    else:
        probs = softmax_rows(hsum.reshape(B * T, H) @ p[sm + ".W"] + p[sm + ".b"]).reshape(B, T, -1)
    cache["probs"] = probs
    return (probs, cache) if want_cache else probs


def cross_entropy_loss(probs, y, total=None):
    """lasagne.objectives.categorical_crossentropy(pred, targets).mean() on (B,C) probabilities
    (avletters/trimodal.py:327-328); `total` overrides the batch size (data parallel)."""
    B = probs.shape[0]
    return -np.log(probs[np.arange(B), y]).sum() / (B if total is None else total)


def bn_running_update(spec, p, cache):
    """The default updates Theano applies with every non-deterministic pass: running mean / inv_std <- (1 - alpha) old +
    alpha batch (lasagne BatchNormLayer.get_output_for, batch_norm_update_averages)."""
    for s, sc in zip(spec["streams"], cache["streams"]):
        if s.get("batchnorm") and cache["training"]:
            bn, dt = s["batchnorm"], p[s["batchnorm"] + ".mean"].dtype.type
            p[bn + ".mean"] = (dt(1) - dt(BN_ALPHA)) * p[bn + ".mean"] + dt(BN_ALPHA) * sc["bn"]["mean"].astype(dt)
            p[bn + ".inv_std"] = (dt(1) - dt(BN_ALPHA)) * p[bn + ".inv_std"] + dt(BN_ALPHA) * sc["bn"]["inv_std"].astype(dt)


def loss_and_grads(spec, p, inputs, targets, mask, theta, total_frames=None, dropout=None, training=None):
    """targets (B,T) int (label repeated over T, runners/3stream.py:360-361)."""
    B, T = mask.shape
    H = spec["lstm_size"]
    probs, cache = forward(spec, p, inputs, mask, theta, want_cache=True, dropout=dropout, training=training)
    g = {}
    sm = spec["softmax_name"]
    if spec.get("head", "frames") == "last":
        y = np.asarray(targets)[:, 0]
        loss = cross_entropy_loss(probs, y, total_frames)
        dz = probs.copy()
        dz[np.arange(B), y] -= 1
        dz = dz / (B if total_frames is None else total_frames)     # softmax + cross-entropy
        hl = cache["hsum"][:, T - 1, :]
        g[sm + ".W"] = hl.T @ dz
        g[sm + ".b"] = dz.sum(0)
        dhsum = np.zeros((B, T, H), dtype=dz.dtype)
        dhsum[:, T - 1, :] = dz @ p[sm + ".W"].T
    else:
        loss = temporal_softmax_loss(probs, targets, mask, total_frames)
        dp = temporal_softmax_loss_bwd(probs, targets, mask, total_frames).reshape(B * T, -1)
        pf = probs.reshape(B * T, -1)
        dz = pf * (dp - (dp * pf).sum(1, keepdims=True))       # through the network's own softmax
        hs = cache["hsum"].reshape(B * T, H)
        g[sm + ".W"] = hs.T @ dz
        g[sm + ".b"] = dz.sum(0)
        dhsum = (dz @ p[sm + ".W"].T).reshape(B, T, H)
    if spec["agg_names"]:
        dfused = 0
        for lc in cache["agg"]:
            dfused = dfused + lstm_bwd(dhsum, lc, p, g)
    else:
        dfused = dhsum
    dfused = dfused * cache["fused_drop"]
    S = len(spec["streams"])
    for k, (s, sc) in enumerate(zip(spec["streams"], cache["streams"])):
        if spec["fusion"] == "concat":
            Hs = int(spec.get("stream_lstm_size") or H)
            dh = dfused[..., k * Hs:(k + 1) * Hs]
        elif spec["fusion"] == "adasum":
            an = "%s.adacoeff%d" % (spec["fuse_name"], k)
            g[an] = np.asarray((dfused * sc["h"]).sum(), dtype=dfused.dtype)
            dh = dfused * p[an]
        else:
            dh = dfused
        dfeat = 0
        for lc in sc["lstm"]:
            dfeat = dfeat + lstm_bwd(dh, lc, p, g)
        dfeat = dfeat * sc["drop"]
        if s.get("aux_dim"):
            dfeat = dfeat[..., :dfeat.shape[-1] - int(s["aux_dim"])]      # the auxiliary input is data
        if s["delta"]:
            dfeat = delta_append_bwd(dfeat, theta)
        da = dfeat.reshape(B * T, -1)
        if s.get("batchnorm"):
            bn, c = s["batchnorm"], sc["bn"]
            g[bn + ".beta"] = da.sum(0)
            g[bn + ".gamma"] = (da * c["xhat"]).sum(0)
            g[bn + ".mean"] = np.zeros_like(p[bn + ".mean"])               # running averages: not trainable
            g[bn + ".inv_std"] = np.zeros_like(p[bn + ".inv_std"])
            gi = p[bn + ".gamma"] * c["inv_std"]
            if cache["training"]:                                          # statistics are functions of the batch
                n_rows = da.shape[0]
                da = gi * (da - g[bn + ".beta"] / n_rows - c["xhat"] * g[bn + ".gamma"] / n_rows)
            else:
                da = gi * da
        for li in range(len(s["enc_names"]) - 1, -1, -1):
            n, act = s["enc_names"][li], s["enc_acts"][li]
            dzl = act_bwd(act, sc["acts"][li + 1], da, sc["kinks"][li], spec.get("relu_grad_at_zero", 0.0))
            g[n + ".W"] = sc["acts"][li].T @ dzl
            g[n + ".b"] = dzl.sum(0)
            da = dzl @ p[n + ".W"].T
    return loss, g, cache


def adam_init(p):
    return dict(t=0, m={k: np.zeros_like(v) for k, v in p.items()},
                v={k: np.zeros_like(v) for k, v in p.items()})


def adam_step(p, g, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """lasagne.updates.adam == custom/updates.py:73-99 with one learning rate."""
    state["t"] += 1
    t = state["t"]
    for k in p:
        dt = p[k].dtype.type
        a_t = dt(lr) * np.sqrt(dt(1) - dt(beta2) ** dt(t)) / (dt(1) - dt(beta1) ** dt(t))
        # (one - beta) is formed in the working precision, as Theano's float32 graph does
        state["m"][k] = dt(beta1) * state["m"][k] + (dt(1) - dt(beta1)) * g[k]
        state["v"][k] = dt(beta2) * state["v"][k] + (dt(1) - dt(beta2)) * g[k] * g[k]
        p[k] = p[k] - a_t * state["m"][k] / (np.sqrt(state["v"][k]) + dt(eps))
    return p


def sgd_step(p, g, lr):
    """lasagne.updates.sgd [upstream]: p -= lr * g."""
    for k in p:
        p[k] = p[k] - p[k].dtype.type(lr) * g[k]
    return p


def momentum_init(p):
    return {k: np.zeros_like(v) for k, v in p.items()}


def momentum_step(p, g, vel, lr, momentum=0.9, nesterov=False):
    """lasagne.updates.momentum / nesterov_momentum [upstream] (avletters/bimodal.py:446-455):
    v = mu v - lr g;  p += v   (classic)      p += mu v - lr g   (Nesterov, with the NEW v)."""
    for k in p:
        dt = p[k].dtype.type
        vel[k] = dt(momentum) * vel[k] - dt(lr) * g[k]
        p[k] = p[k] + (dt(momentum) * vel[k] - dt(lr) * g[k] if nesterov else vel[k])
    return p


def adadelta_init(p):
    return dict(accu={k: np.zeros_like(v) for k, v in p.items()}, delta={k: np.zeros_like(v) for k, v in p.items()})


def adadelta_step(p, g, state, lr=1.0, rho=0.95, eps=1e-6):
    """lasagne.updates.adadelta [upstream] (avletters/avletters_convae.py:230 uses lr 0.8, rho default)."""
    for k in p:
        dt = p[k].dtype.type
        state["accu"][k] = dt(rho) * state["accu"][k] + (dt(1) - dt(rho)) * g[k] * g[k]
        upd = g[k] * np.sqrt(state["delta"][k] + dt(eps)) / np.sqrt(state["accu"][k] + dt(eps))
        p[k] = p[k] - dt(lr) * upd
        state["delta"][k] = dt(rho) * state["delta"][k] + (dt(1) - dt(rho)) * upd * upd
    return p


def train_step(spec, p, state, inputs, targets, mask, theta, lr, training=False):
    """training: BatchNorm layers in batch-statistics mode incl. the running-average update (what ``train`` of the
    reference scripts does: get_output(network, deterministic=False)); dropout stays off here (deterministic masks are
    a separate argument of loss_and_grads)."""
    loss, g, cache = loss_and_grads(spec, p, inputs, targets, mask, theta, training=training)
    adam_step(p, g, state, lr)
    bn_running_update(spec, p, cache)
    return loss


def majority_vote(probs, mask):
    """evaluate_model2 (runners/3stream.py:48-82): argmax over the first len_i frames,
    vote histogram, argmax (ties -> lowest class id)."""
    lens = mask.sum(-1).astype(int)
    C = probs.shape[-1]
    out = np.zeros(len(probs), int)
    for i, eg in enumerate(probs):
        pred = np.argmax(eg[:lens[i]], axis=-1)
        out[i] = np.argmax(np.bincount(pred, minlength=C))
    return out


# --------------------------------------------------------------------------- #
# convenience spec builders mirroring the model zoo
# --------------------------------------------------------------------------- #
ENC_NAMES = ["fc1", "fc2", "fc3", "bottleneck"]


def spec_nstream(input_dims, enc_shapes=(2000, 1000, 500, 50),
                 enc_acts=("rectify", "rectify", "rectify", "linear"),
                 lstm_size=250, classes=26, fusion="concat", peepholes=False,
                 has_encoder=None, delta=None):
    """adenet_{2,3,4}stream.create_model-shaped spec (modelzoo/adenet_3stream.py:145-262)."""
    S = len(input_dims)
    has_encoder = has_encoder or [True] * S
    delta = delta or [True] * S
    streams = []
    for k, d in enumerate(input_dims):
        sfx = "_s%d" % (k + 1)
        enc = has_encoder[k]
        streams.append(dict(input_dim=d,
                            enc_names=[n + sfx for n in ENC_NAMES[:len(enc_shapes)]] if enc else [],
                            enc_shapes=list(enc_shapes) if enc else [],
                            enc_acts=list(enc_acts) if enc else [],
                            delta=delta[k], lstm_names=["lstm" + sfx], peepholes=peepholes))
    return dict(streams=streams, fusion=fusion, fuse_name={"adasum": "adasum1", "sum": "sum1",
                                                           "concat": "concat"}[fusion],
                agg_names=["f_lstm_agg", "b_lstm_agg"], agg_peepholes=False,
                lstm_size=lstm_size, classes=classes, softmax_name="softmax")


def spec_deltanet(input_dim, enc_shapes=(2000, 1000, 500, 50),
                  enc_acts=("rectify", "rectify", "rectify", "linear"),
                  lstm_size=250, classes=26, peepholes=False, use_blstm=True):
    """deltanet_majority_vote.create_model (modelzoo/deltanet_majority_vote.py:14-66)."""
    names = ["f_blstm1", "b_blstm1"] if use_blstm else ["lstm"]
    return dict(streams=[dict(input_dim=input_dim, enc_names=ENC_NAMES[:len(enc_shapes)],
                              enc_shapes=list(enc_shapes), enc_acts=list(enc_acts), delta=True,
                              lstm_names=names, peepholes=peepholes)],
                fusion="none", fuse_name="", agg_names=[], agg_peepholes=False,
                lstm_size=lstm_size, classes=classes, softmax_name="softmax")


def spec_adenet_v3(raw_dim, dct_dim, diff_dim, enc_shapes=(2000, 1000, 500, 50),
                   enc_acts=("rectify", "rectify", "rectify", "linear"), lstm_size=250, classes=26, fusion="concat"):
    """modelzoo/adenet_v3.create_model (:64-188): raw + diff encoder streams with deltas, a raw DCT stream, dropout
    0.5 / 0.2 / 0.5 ahead of the three LSTMs of 2*lstm_size units, dropout 0.5 on the fused tensor, summed BLSTM of
    2*lstm_size units, LAST time step, softmax; trained with categorical cross-entropy.  No LSTMLayer of that file
    passes ``peepholes=``, so Lasagne's default True applies to all five (:27-45, :113-143)."""
    enc = lambda sfx: dict(enc_names=[n + sfx for n in ENC_NAMES[:len(enc_shapes)]], enc_shapes=list(enc_shapes),
                           enc_acts=list(enc_acts), delta=True)
    streams = [dict(input_dim=raw_dim, lstm_names=["lstm_raw"], peepholes=True, dropout=0.5, **enc("_raw")),
               dict(input_dim=dct_dim, enc_names=[], enc_shapes=[], enc_acts=[], delta=False, lstm_names=["lstm_dct"],
                    peepholes=True, dropout=0.2),
               dict(input_dim=diff_dim, lstm_names=["lstm_diff"], peepholes=True, dropout=0.5, **enc("_diff"))]
    return dict(streams=streams, fusion=fusion, fuse_name={"adasum": "adasum1", "sum": "sum1", "concat": "concat"}[fusion],
                agg_names=["f_lstm_agg", "b_lstm_agg"], agg_peepholes=True, agg_dropout=0.5,
                lstm_size=2 * lstm_size, classes=classes, softmax_name="output", head="last", loss="cross_entropy")


def spec_adenet_v1(input_dim, dct_dim, enc_shapes=(2000, 1000, 500, 50), enc_acts=("sigmoid", "sigmoid", "sigmoid", "linear"),
                   lstm_size=250, classes=26, v1_1=False):
    """modelzoo/adenet_v1.create_model (:48-109) / adenet_v1_1.create_model: encoder 'fc1'..'bottleneck' -> BatchNormLayer
    'batchnorm1' -> DeltaLayer -> ConcatLayer with the DCT input -> [v1_1: dropout 0.5] -> summed BLSTM 'f_lstm1' /
    'b_lstm1' of lstm_size (v1_1: 2 * lstm_size) units -> [v1_1: dropout 0.5] -> summed BLSTM 'f_lstm2' / 'b_lstm2' of
    2 * lstm_size units -> LAST time step -> softmax 'output'.  No LSTMLayer passes ``peepholes=``: Lasagne's default
    True applies (:27-43)."""
    stream = dict(input_dim=input_dim, enc_names=ENC_NAMES[:len(enc_shapes)], enc_shapes=list(enc_shapes), enc_acts=list(enc_acts),
                  delta=True, batchnorm="batchnorm1", aux_dim=dct_dim, lstm_names=["f_lstm1", "b_lstm1"], peepholes=True,
                  dropout=0.5 if v1_1 else 0.0)
    return dict(streams=[stream], fusion="none", fuse_name="", agg_names=["f_lstm2", "b_lstm2"], agg_peepholes=True,
                agg_dropout=0.5 if v1_1 else 0.0, lstm_size=2 * lstm_size, stream_lstm_size=(2 * lstm_size if v1_1 else lstm_size),
                classes=classes, softmax_name="output", head="last", loss="cross_entropy")


def spec_deltanet_last(input_dim, enc_shapes=(2000, 1000, 500, 50), enc_acts=("rectify", "rectify", "rectify", "linear"),
                       lstm_size=250, classes=26):
    """modelzoo/deltanet.create_model (:12-56): encoder + deltas + summed BLSTM, LAST time step, softmax."""
    s = spec_deltanet(input_dim, enc_shapes, enc_acts, lstm_size, classes, peepholes=False, use_blstm=True)
    s.update(head="last", loss="cross_entropy", softmax_name="output")
    return s
