"""Pins the CPU oracle itself (no GPU): finite differences, torch.nn.LSTM as a second
opinion, delta-layer known answers ('derived')."""
import numpy as np
import pytest
import torch

from oracle import adenet_oracle as O


def ragged_mask(rng, B, T, full_first=True):
    lens = rng.integers(2, T + 1, size=B)
    if full_first:
        lens[0] = T
    m = np.zeros((B, T), np.uint8)
    for i, l in enumerate(lens):
        m[i, :l] = 1
    return m


def small_spec(fusion, peep, S=2):
    spec = O.spec_nstream([7, 5, 6][:S], enc_shapes=(6, 4), enc_acts=("rectify", "linear"),
                          lstm_size=5, classes=4, fusion=fusion, peepholes=peep)
    return spec


# ----------------------------------------------------------------------------- delta
def test_delta_closed_form_matches_literal_scans():
    rng = np.random.default_rng(0)
    for theta in (1, 3, 9):
        A = rng.normal(size=(12, 5)).astype(np.float32)
        lit = O.append_delta_literal(A, theta)
        cf = O.delta_append(A[None], theta)[0]
        assert np.abs(lit - cf).max() < 5e-5


def test_delta_derived_known_answers():
    # array printed by utils/signal.py:95-100 at theta=1; answers derived in SURVEY.md §8c
    seqs = np.array([[[1, 2, 3, 4, 5], [10, 12, 13, 14, 15], [300, 1, 23, 56, 22]],
                     [[1, 1, 1, 1, 1], [1, 1, 100, 1, 1], [1, 1, 1, 1, 1]]], dtype=np.float32)
    out = O.delta_append(seqs, 1)
    np.testing.assert_allclose(out[0, 0], [1, 2, 3, 4, 5, 4.5, 5, 5, 5, 5, 72.5, -2.75, 2.5, 10.5, 1.75])
    np.testing.assert_allclose(out[0, 1], [10, 12, 13, 14, 15, 149.5, -0.5, 10, 26, 8.5,
                                           70.25, -5.25, 0, 8, -0.75])
    np.testing.assert_allclose(out[0, 2], [300, 1, 23, 56, 22, 145, -5.5, 5, 21, 3.5,
                                           -2.25, -2.5, -2.5, -2.5, -2.5])
    np.testing.assert_allclose(out[1, 0, 5:10], [0, 0, 49.5, 0, 0])
    np.testing.assert_allclose(out[1, 0, 10:], [0, 0, -24.75, 0, 0])
    for b in range(2):
        np.testing.assert_allclose(O.append_delta_literal(seqs[b], 1), out[b])


def test_delta_adjoint():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(3, 9, 4))
    g = rng.normal(size=(3, 9, 12))
    lhs = (O.delta_append(x, 3) * g).sum()
    rhs = (x * O.delta_append_bwd(g, 3)).sum()
    assert abs(lhs - rhs) < 1e-10


# ----------------------------------------------------------------------------- LSTM vs torch
@pytest.mark.parametrize("backwards", [False, True])
def test_lstm_matches_torch(backwards):
    rng = np.random.default_rng(2)
    B, T, F, H = 3, 6, 4, 5
    spec = dict(lstm_size=H)
    p = {}
    for g in O.GATES:
        p["l.W_in_to_" + g] = rng.normal(0, .5, (F, H))
        p["l.W_hid_to_" + g] = rng.normal(0, .5, (H, H))
        p["l.b_" + g] = rng.normal(0, .5, (H,))
    p["l.hid_init"] = np.zeros((1, H))
    p["l.cell_init"] = np.zeros((1, H))
    x = rng.normal(size=(B, T, F))
    mask = np.ones((B, T), np.uint8)
    hs, cache = O.lstm_fwd(x, mask, p, "l", backwards=backwards)
    dhs = rng.normal(size=hs.shape) * 0.1        # small: the +-5 clip must not fire vs torch
    grads = {}
    dx = O.lstm_bwd(dhs, cache, p, grads)

    lstm = torch.nn.LSTM(F, H, batch_first=True).double()
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.tensor(np.concatenate([p["l.W_in_to_" + g] for g in O.GATES], 1).T))
        lstm.weight_hh_l0.copy_(torch.tensor(np.concatenate([p["l.W_hid_to_" + g] for g in O.GATES], 1).T))
        lstm.bias_ih_l0.copy_(torch.tensor(np.concatenate([p["l.b_" + g] for g in O.GATES])))
        lstm.bias_hh_l0.zero_()
    xt = torch.tensor(x[:, ::-1].copy() if backwards else x, requires_grad=True)
    out, _ = lstm(xt)
    ref = out.detach().numpy()
    if backwards:
        ref = ref[:, ::-1]
    np.testing.assert_allclose(hs, ref, atol=1e-12)
    gout = torch.tensor(dhs[:, ::-1].copy() if backwards else dhs)
    out.backward(gout)
    dx_ref = xt.grad.numpy()
    if backwards:
        dx_ref = dx_ref[:, ::-1]
    np.testing.assert_allclose(dx, dx_ref, atol=1e-12)
    dWhh = lstm.weight_hh_l0.grad.numpy().T
    np.testing.assert_allclose(np.concatenate([grads["l.W_hid_to_" + g] for g in O.GATES], 1), dWhh, atol=1e-12)


def test_lstm_mask_holds_state_and_backward_starts_in_padding():
    rng = np.random.default_rng(3)
    spec = O.spec_deltanet(4, enc_shapes=(), enc_acts=(), lstm_size=3, classes=2)
    p = O.init_params(spec, rng, np.float64, perturb=0.3)
    x = rng.normal(size=(2, 5, 12))
    mask = np.array([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], np.uint8)
    hf, _ = O.lstm_fwd(x, mask, p, "f_blstm1")
    hb, _ = O.lstm_fwd(x, mask, p, "b_blstm1", backwards=True)
    np.testing.assert_array_equal(hf[0, 3], hf[0, 2])          # held on padded steps
    np.testing.assert_array_equal(hf[0, 4], hf[0, 2])
    np.testing.assert_allclose(hb[0, 4], p["b_blstm1.hid_init"][0])   # still at init inside padding
    # a short utterance equals the same utterance run alone at its own length (forward dir.)
    h_alone, _ = O.lstm_fwd(x[:1, :3], np.ones((1, 3), np.uint8), p, "f_blstm1")
    np.testing.assert_allclose(hf[0, :3], h_alone[0], atol=1e-14)


# ----------------------------------------------------------------------------- finite differences
@pytest.mark.parametrize("fusion,peep,S", [("concat", False, 2), ("sum", True, 2), ("adasum", True, 3)])
def test_gradients_finite_difference(fusion, peep, S):
    rng = np.random.default_rng(4)
    spec = small_spec(fusion, peep, S)
    p = O.init_params(spec, rng, np.float64, enc_std=0.5, perturb=0.2)
    B, T = 3, 6
    mask = ragged_mask(rng, B, T)
    inputs = [rng.normal(size=(B, T, s["input_dim"])) * mask[..., None] for s in spec["streams"]]
    y = np.repeat(rng.integers(0, spec["classes"], size=(B, 1)), T, axis=1)
    loss, g, _ = O.loss_and_grads(spec, p, inputs, y, mask, 2)
    assert set(g) == set(O.param_names(spec))
    eps = 1e-6
    for name in O.param_names(spec):
        flat = p[name].reshape(-1)
        for j in rng.choice(flat.size, size=min(3, flat.size), replace=False):
            old = flat[j]
            flat[j] = old + eps
            lp = O.temporal_softmax_loss(O.forward(spec, p, inputs, mask, 2), y, mask)
            flat[j] = old - eps
            lm = O.temporal_softmax_loss(O.forward(spec, p, inputs, mask, 2), y, mask)
            flat[j] = old
            num = (lp - lm) / (2 * eps)
            ana = np.asarray(g[name]).reshape(-1)[j]
            assert abs(num - ana) < 1e-7 + 1e-5 * abs(num), (name, j, num, ana)


def test_deltanet_blstm_finite_difference():
    rng = np.random.default_rng(5)
    spec = O.spec_deltanet(6, enc_shapes=(5, 3), enc_acts=("sigmoid", "linear"), lstm_size=4, classes=3,
                           peepholes=True, use_blstm=True)
    p = O.init_params(spec, rng, np.float64, enc_std=0.5, perturb=0.2)
    B, T = 2, 5
    mask = ragged_mask(rng, B, T)
    inputs = [rng.normal(size=(B, T, 6))]
    y = np.repeat(rng.integers(0, 3, size=(B, 1)), T, axis=1)
    loss, g, _ = O.loss_and_grads(spec, p, inputs, y, mask, 3)
    eps = 1e-6
    for name in O.param_names(spec):
        flat = p[name].reshape(-1)
        j = int(rng.integers(flat.size))
        old = flat[j]
        flat[j] = old + eps
        lp = O.temporal_softmax_loss(O.forward(spec, p, inputs, mask, 3), y, mask)
        flat[j] = old - eps
        lm = O.temporal_softmax_loss(O.forward(spec, p, inputs, mask, 3), y, mask)
        flat[j] = old
        num = (lp - lm) / (2 * eps)
        assert abs(num - np.asarray(g[name]).reshape(-1)[j]) < 1e-7 + 1e-5 * abs(num), name


def test_grad_clip_fires():
    """With huge upstream gradients the per-step gate gradient is clamped to +-5."""
    rng = np.random.default_rng(6)
    spec = O.spec_deltanet(4, enc_shapes=(), enc_acts=(), lstm_size=3, classes=2, use_blstm=False)
    p = O.init_params(spec, rng, np.float64, perturb=0.3)
    x = rng.normal(size=(2, 4, 12))
    mask = np.ones((2, 4), np.uint8)
    hs, cache = O.lstm_fwd(x, mask, p, "lstm")
    g = {}
    O.lstm_bwd(np.full_like(hs, 1e4), cache, p, g)
    # db = sum over (B,T) of clipped dgates -> bounded by 5*B*T
    for gate in O.GATES:
        assert np.abs(g["lstm.b_" + gate]).max() <= 5 * 2 * 4 + 1e-9
    assert max(np.abs(g["lstm.b_" + gate]).max() for gate in O.GATES) > 5  # and it did saturate


# ----------------------------------------------------------------------------- loss / adam / vote
def test_double_softmax_loss_floor():
    C = 26
    probs = np.zeros((1, 3, C)); probs[..., 0] = 1.0
    y = np.zeros((1, 3), int)
    mask = np.ones((1, 3), np.uint8)
    loss = O.temporal_softmax_loss(probs, y, mask)
    np.testing.assert_allclose(loss, np.log(1 + (C - 1) / np.e))


def test_loss_normaliser_is_valid_frames():
    rng = np.random.default_rng(7)
    probs = O.softmax_rows(rng.normal(size=(2, 4, 3)))
    y = rng.integers(0, 3, size=(2, 4))
    mask = np.array([[1, 1, 0, 0], [1, 1, 1, 1]], np.uint8)
    q = O.softmax_rows(probs)
    want = -sum(np.log(q[b, t, y[b, t]]) for b in range(2) for t in range(4) if mask[b, t]) / 6
    np.testing.assert_allclose(O.temporal_softmax_loss(probs, y, mask), want)


def test_adam_matches_formula():
    p = {"w": np.array([1.0, -2.0])}
    g = {"w": np.array([0.5, -0.25])}
    st = O.adam_init(p)
    O.adam_step(p, g, st, lr=0.1)
    # first step of Adam moves every coordinate by ~lr*sign(g)
    np.testing.assert_allclose(p["w"], [1.0 - 0.1, -2.0 + 0.1], atol=1e-6)
    assert st["t"] == 1


def test_majority_vote_ties_lowest_class():
    probs = np.zeros((1, 4, 3))
    probs[0, 0, 2] = 1; probs[0, 1, 1] = 1; probs[0, 2, 2] = 1; probs[0, 3, 1] = 1
    mask = np.ones((1, 4), np.uint8)
    assert O.majority_vote(probs, mask)[0] == 1
    mask[0, 3] = 0
    assert O.majority_vote(probs, mask)[0] == 2


def test_param_order_follows_lasagne_topology():
    spec = O.spec_nstream([8, 8, 8], enc_shapes=(4, 2), enc_acts=("rectify", "linear"), lstm_size=3,
                          classes=2, fusion="adasum", peepholes=True)
    names = O.param_names(spec)
    assert names[:4] == ["fc1_s1.W", "fc1_s1.b", "fc2_s1.W", "fc2_s1.b"]
    assert names[4:7] == ["lstm_s1.W_in_to_ingate", "lstm_s1.W_hid_to_ingate", "lstm_s1.b_ingate"]
    i = names.index("lstm_s1.W_cell_to_ingate")
    assert names[i:i + 5] == ["lstm_s1.W_cell_to_ingate", "lstm_s1.W_cell_to_forgetgate",
                              "lstm_s1.W_cell_to_outgate", "lstm_s1.cell_init", "lstm_s1.hid_init"]
    assert names.index("adasum1.adacoeff0") < names.index("f_lstm_agg.W_in_to_ingate")
    assert "f_lstm_agg.W_cell_to_ingate" not in names          # create_blstm default use_peepholes=False
    assert names[-2:] == ["softmax.W", "softmax.b"]
    n = sum(int(np.prod(s)) for s in O.param_shapes(spec).values())
    assert n == sum(int(np.prod(O.param_shapes(spec)[k])) for k in names)


def test_param_budget_matches_survey_appendix_d():
    spec = O.spec_nstream([1200, 1200, 1200])
    shapes = O.param_shapes(spec)
    assert sum(int(np.prod(shapes[k])) for k in O.param_names(spec)) == 17999676


# --------------------------------------------------------------------------- SURVEY 8f-1 additions
def test_last_timestep_head_cross_entropy_and_dropout_backward_by_finite_differences():
    """adenet_v3-shaped graph in float64 with the stochastic layers ON (fixed hash masks): every analytic gradient of
    oracle.loss_and_grads against central differences; and the head really reads the LAST padded row (App. E-3)."""
    spec = O.spec_adenet_v3(11, 6, 11, enc_shapes=(8, 4), enc_acts=("sigmoid", "linear"), lstm_size=3, classes=4)
    rng = np.random.default_rng(21)
    p = O.init_params(spec, rng, np.float64, enc_std=0.4, perturb=0.2)
    B, T, theta = 4, 6, 2
    mask = np.ones((B, T), np.uint8); mask[1, 4:] = 0; mask[3, 1:] = 0
    inputs = [rng.normal(size=(B, T, s["input_dim"])) * mask[..., None] for s in spec["streams"]]
    y = np.repeat(rng.integers(0, 4, size=(B, 1)), T, axis=1)
    dr = dict(seed=31337, counter=2)
    loss, g, cache = O.loss_and_grads(spec, p, inputs, y, mask, theta, dropout=dr)
    probs = cache["probs"]
    assert probs.shape == (B, 4) and np.allclose(probs.sum(1), 1)
    assert np.isclose(loss, -np.log(probs[np.arange(B), y[:, 0]]).mean())
    for k in O.param_names(spec):
        for _ in range(2):
            idx = tuple(rng.integers(0, n) for n in p[k].shape)
            q = {a: b.copy() for a, b in p.items()}
            q[k][idx] += 1e-6
            lp = O.loss_and_grads(spec, q, inputs, y, mask, theta, dropout=dr)[0]
            q[k][idx] -= 2e-6
            lm = O.loss_and_grads(spec, q, inputs, y, mask, theta, dropout=dr)[0]
            fd = (lp - lm) / 2e-6
            assert abs(fd - g[k][idx]) <= 1e-6 + 1e-5 * abs(fd), (k, idx, fd, g[k][idx])
    # deterministic pass = no scaling anywhere; masks keep ~ (1 - p) of the elements and rescale by 1 / (1 - p)
    det = O.forward(spec, p, inputs, mask, theta)
    assert not np.allclose(det, probs)
    sc = O.dropout_scale((64, 50, 20), 0.2, dict(seed=1, counter=0), 3, np.float32)
    assert set(np.unique(sc)) == {np.float32(0), np.float32(1.25)} and abs((sc > 0).mean() - 0.8) < 0.01
    assert not np.array_equal(sc, O.dropout_scale((64, 50, 20), 0.2, dict(seed=1, counter=1), 3, np.float32))


def test_update_rules_follow_the_lasagne_formulas():
    """Two hand-computed steps of momentum / nesterov / adadelta on a scalar."""
    p, g = {"w": np.array([1.0])}, {"w": np.array([0.5])}
    v = O.momentum_init(p)
    O.momentum_step(p, g, v, 0.1, 0.9)
    assert np.isclose(p["w"][0], 0.95) and np.isclose(v["w"][0], -0.05)
    O.momentum_step(p, g, v, 0.1, 0.9)
    assert np.isclose(v["w"][0], -0.095) and np.isclose(p["w"][0], 0.855)
    p, v = {"w": np.array([1.0])}, {"w": np.array([0.0])}
    O.momentum_step(p, g, v, 0.1, 0.9, nesterov=True)
    assert np.isclose(p["w"][0], 1 + 0.9 * -0.05 - 0.05)
    p, st = {"w": np.array([1.0])}, O.adadelta_init({"w": np.array([1.0])})
    O.adadelta_step(p, g, st, 1.0, 0.95, 1e-6)
    accu = 0.05 * 0.25
    upd = 0.5 * np.sqrt(1e-6) / np.sqrt(accu + 1e-6)
    assert np.isclose(p["w"][0], 1 - upd) and np.isclose(st["delta"]["w"][0], 0.05 * upd * upd)
    p = {"w": np.array([1.0])}
    O.sgd_step(p, g, 0.1)
    assert np.isclose(p["w"][0], 0.95)


def test_batchnorm_auxiliary_input_and_two_lstm_widths_by_finite_differences():
    """adenet_v1-shaped graph in float64 (modelzoo/adenet_v1.py:48-109): BatchNormLayer behind the encoder, DCT input
    concatenated after the delta layer, stream BLSTM of lstm_size units under an aggregation BLSTM of 2 * lstm_size.
    Every analytic gradient against central differences, in the training mode (batch statistics: the mean / variance
    terms of the BatchNorm adjoint) and in the deterministic mode (running averages); running-average update rule."""
    spec = O.spec_adenet_v1(9, 5, enc_shapes=(7, 4), enc_acts=("sigmoid", "linear"), lstm_size=3, classes=4)
    shapes = O.param_shapes(spec)
    assert shapes["f_lstm1.W_hid_to_cell"] == (3, 3) and shapes["f_lstm2.W_in_to_ingate"] == (3, 6)
    assert shapes["f_lstm1.W_in_to_ingate"] == (3 * 4 + 5, 3) and shapes["batchnorm1.gamma"] == (4,)
    names = O.param_names(spec)
    i = names.index(spec["streams"][0]["enc_names"][-1] + ".b")
    assert names[i + 1:i + 5] == ["batchnorm1.beta", "batchnorm1.gamma", "batchnorm1.mean", "batchnorm1.inv_std"]
    assert names[i + 5] == "f_lstm1.W_in_to_ingate" and "f_lstm1.W_cell_to_ingate" in names
    rng = np.random.default_rng(5)
    p = O.init_params(spec, rng, np.float64, enc_std=0.4, perturb=0.2)
    p["batchnorm1.inv_std"] = np.abs(p["batchnorm1.inv_std"]) + 0.5
    B, T, theta = 4, 6, 2
    mask = np.ones((B, T), np.uint8); mask[1, 4:] = 0; mask[3, 2:] = 0
    inputs = [rng.normal(size=(B, T, 9)) * mask[..., None], rng.normal(size=(B, T, 5)) * mask[..., None]]
    y = np.repeat(rng.integers(0, 4, size=(B, 1)), T, axis=1)
    for training in (True, False):
        loss, g, cache = O.loss_and_grads(spec, p, inputs, y, mask, theta, training=training)
        for k in names:
            if k.endswith((".mean", ".inv_std")):
                assert not g[k].any()                          # not trainable (in deterministic mode they ARE inputs of the
                continue                                       # graph, but the reference never differentiates them)
            for _ in range(2):
                idx = tuple(rng.integers(0, n) for n in p[k].shape)
                q = {a: b.copy() for a, b in p.items()}
                q[k][idx] += 1e-6
                lp = O.loss_and_grads(spec, q, inputs, y, mask, theta, training=training)[0]
                q[k][idx] -= 2e-6
                lm = O.loss_and_grads(spec, q, inputs, y, mask, theta, training=training)[0]
                fd = (lp - lm) / 2e-6
                assert abs(fd - g[k][idx]) <= 1e-6 + 2e-5 * abs(fd), (training, k, idx, fd, g[k][idx])
    loss, g, cache = O.loss_and_grads(spec, p, inputs, y, mask, theta, training=True)
    bn = cache["streams"][0]["bn"]
    assert np.allclose(bn["xhat"].mean(0), 0, atol=1e-12) and np.allclose((bn["xhat"] ** 2).mean(0), 1, atol=1e-2)     # (eps = 1e-4 under the root)
    old_mean, old_inv = p["batchnorm1.mean"].copy(), p["batchnorm1.inv_std"].copy()
    O.bn_running_update(spec, p, cache)
    assert np.allclose(p["batchnorm1.mean"], 0.9 * old_mean + 0.1 * bn["mean"])
    assert np.allclose(p["batchnorm1.inv_std"], 0.9 * old_inv + 0.1 * bn["inv_std"])
    # torch's batch_norm as an independent implementation of the training-mode forward
    import torch
    a = torch.tensor(bn["x"])
    ref = torch.nn.functional.batch_norm(a, None, None, torch.tensor(p["batchnorm1.gamma"]), torch.tensor(p["batchnorm1.beta"]),
                                         training=True, eps=O.BN_EPS).numpy()
    assert np.allclose(bn["xhat"] * p["batchnorm1.gamma"] + p["batchnorm1.beta"], ref, atol=1e-10)
