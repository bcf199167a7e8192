"""the last-timestep model family -- SliceLayer(-1) head, categorical cross-entropy, DropoutLayers,
sgd / momentum / nesterov / adadelta updates (modelzoo/adenet_v3.py:64-188, deltanet.py:12-56,
avletters/trimodal.py:327-328, avletters/bimodal.py:446-455) -- against the NumPy oracle.

Dropout parity: Theano's MRG stream is not reproducible, so the mask is DEFINED by a counter-based hash shared by
csrc/elementwise.hip and oracle/adenet_oracle.py; with the same (seed, counter) both draw the same mask and the
stochastic passes are compared like the deterministic ones (fp32 mode: 1e-4 relative on every gradient tensor)."""
import numpy as np
import pytest

from oracle import adenet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_cls():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ip_avsr_amd.model import AdeNetModel
    return AdeNetModel


def case(spec, B, T, seed):
    rng = np.random.default_rng(seed)
    p = O.init_params(spec, rng, np.float32, enc_std=0.3, perturb=0.1)
    lens = rng.integers(1, T + 1, size=B); lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    inputs = [(rng.normal(size=(B, T, s["input_dim"])) * mask[..., None]).astype(np.float32) for s in spec["streams"]]
    y = np.repeat(rng.integers(0, spec["classes"], size=(B, 1)), T, axis=1).astype(np.int32)
    return p, inputs, y, mask


def specs():
    return {
        "adenet_v3_concat": O.spec_adenet_v3(12, 7, 12, enc_shapes=(9, 4), enc_acts=("rectify", "linear"), lstm_size=4, classes=5),
        "adenet_v3_sum": O.spec_adenet_v3(10, 6, 10, enc_shapes=(8, 5), enc_acts=("sigmoid", "linear"), lstm_size=3, classes=4,
                                         fusion="sum"),
        "deltanet_last": O.spec_deltanet_last(11, enc_shapes=(13, 5), enc_acts=("rectify", "linear"), lstm_size=6, classes=4),
    }


def check_grads(g, g_ref, spec, tol=1e-4):
    gscale = max(np.abs(v).max() for v in g_ref.values())
    for k in O.param_names(spec):
        err = np.abs(g[k] - g_ref[k]).max()
        assert err <= tol * max(np.abs(g_ref[k]).max(), 1e-3 * gscale) + 1e-9, (k, err, np.abs(g_ref[k]).max())


@pytest.mark.parametrize("name", list(specs()))
def test_last_timestep_head_deterministic_and_stochastic(model_cls, name):
    spec = specs()[name]
    B, T, theta = 6, 8, 2
    p, inputs, y, mask = case(spec, B, T, seed=sum(map(ord, name)))
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    in64 = [x.astype(np.float64) for x in inputs]
    m = model_cls(spec)
    m.set_params_dict(p)
    probs = m.predict(inputs, mask, theta)
    assert probs.shape == (B, spec["classes"])
    assert np.abs(probs - O.forward(spec, p64, in64, mask, theta)).max() <= 2e-5
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, in64, y, mask, theta)
    assert abs(m.loss(inputs, y, mask, theta) - l_ref) <= 1e-5 * abs(l_ref)
    l = m.compute_grads(inputs, y, mask, theta, deterministic=True)
    assert abs(l - l_ref) <= 1e-5 * abs(l_ref)
    check_grads(m.get_grads_dict(), g_ref, spec)
    # stochastic: the same masks on both sides
    for counter in (0, 7):
        dr = dict(seed=20240917, counter=counter)
        l_ref, g_ref, _ = O.loss_and_grads(spec, p64, in64, y, mask, theta, dropout=dr)
        m.set_dropout_state(dr["seed"], counter)
        assert abs(m.loss(inputs, y, mask, theta, deterministic=False) - l_ref) <= 1e-5 * abs(l_ref)
        m.set_dropout_state(dr["seed"], counter)
        l = m.compute_grads(inputs, y, mask, theta)
        assert abs(l - l_ref) <= 1e-5 * abs(l_ref)
        check_grads(m.get_grads_dict(), g_ref, spec)
    # the counter advances: two stochastic passes differ, deterministic ones do not
    a = m.loss(inputs, y, mask, theta, deterministic=False)
    b = m.loss(inputs, y, mask, theta, deterministic=False)
    has_dropout = bool(spec.get("agg_dropout")) or any(s.get("dropout") for s in spec["streams"])
    assert (a != b) == has_dropout
    assert m.loss(inputs, y, mask, theta) == m.loss(inputs, y, mask, theta)
    m.close()


def test_dropout_on_the_per_frame_family_too(model_cls):
    """adenet_3stream_dropout-style: per-frame head and temporal loss with dropout ahead of the LSTMs and on the concat."""
    spec = O.spec_nstream([12, 9], enc_shapes=(14, 6), enc_acts=("rectify", "linear"), lstm_size=7, classes=4, fusion="concat")
    for s in spec["streams"]:
        s["dropout"] = 0.3
    spec["agg_dropout"] = 0.4
    B, T, theta = 5, 9, 3
    p, inputs, y, mask = case(spec, B, T, seed=5)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    in64 = [x.astype(np.float64) for x in inputs]
    m = model_cls(spec)
    m.set_params_dict(p)
    dr = dict(seed=99, counter=3)
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, in64, y, mask, theta, dropout=dr)
    m.set_dropout_state(99, 3)
    l = m.compute_grads(inputs, y, mask, theta)
    assert abs(l - l_ref) <= 1e-5 * abs(l_ref)
    check_grads(m.get_grads_dict(), g_ref, spec)
    assert np.abs(m.predict(inputs, mask, theta) - O.forward(spec, p64, in64, mask, theta)).max() <= 2e-5
    m.close()


@pytest.mark.parametrize("rule", ["sgd", "momentum", "nesterov_momentum", "adadelta"])
def test_update_rules_match_lasagne_formulas(model_cls, rule):
    spec = specs()["deltanet_last"]
    B, T, theta = 5, 7, 2
    p, inputs, y, mask = case(spec, B, T, seed=11)
    m = model_cls(spec)
    m.set_params_dict(p)
    ref = {k: v.copy() for k, v in p.items()}
    vel, ad = O.momentum_init(ref), O.adadelta_init(ref)
    for step in range(3):
        _, g, _ = O.loss_and_grads(spec, ref, inputs, y, mask, theta)
        g = {k: np.asarray(v, np.float32) for k, v in g.items()}
        m.compute_grads(inputs, y, mask, theta, deterministic=True)
        if rule == "sgd":
            O.sgd_step(ref, g, 0.05); m.apply_sgd(0.05)
        elif rule == "momentum":
            O.momentum_step(ref, g, vel, 0.05, 0.9); m.apply_sgd(0.05, 0.9)
        elif rule == "nesterov_momentum":
            O.momentum_step(ref, g, vel, 0.05, 0.9, nesterov=True); m.apply_sgd(0.05, 0.9, nesterov=True)
        else:
            O.adadelta_step(ref, g, ad, 0.8, 0.95, 1e-6); m.apply_adadelta(0.8, 0.95, 1e-6)
    got = m.get_params_dict()
    for k in ref:
        assert np.abs(got[k] - ref[k]).max() <= 2e-4 * max(np.abs(ref[k]).max(), 1e-2), (k, np.abs(got[k] - ref[k]).max())
    m.close()


def test_last_head_bf16_tracks_the_oracle_with_and_without_the_fused_concat(model_cls, monkeypatch):
    """bf16 mode runs the weight-stationary LSTM kernels and, for concat, the one-GEMM aggregation input with the
    dropped stream outputs; same masks as the oracle."""
    spec = dict(specs()["adenet_v3_concat"], precision="bf16")
    B, T, theta = 9, 8, 2
    p, inputs, y, mask = case(spec, B, T, seed=3)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    in64 = [x.astype(np.float64) for x in inputs]
    dr = dict(seed=7, counter=1)
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, in64, y, mask, theta, dropout=dr)
    for no_cat in (False, True):
        if no_cat:
            monkeypatch.setenv("ADN_NO_CAT", "1")
        m = model_cls(spec)
        m.set_params_dict(p)
        assert np.abs(m.predict(inputs, mask, theta) - O.forward(spec, p64, in64, mask, theta)).max() <= 2e-2
        m.set_dropout_state(7, 1)
        l = m.compute_grads(inputs, y, mask, theta)
        assert abs(l - l_ref) <= 2e-2 * abs(l_ref)
        g = m.get_grads_dict()
        for k in O.param_names(spec):
            a, b = np.asarray(g_ref[k], np.float64).ravel(), g[k].ravel().astype(np.float64)
            if np.linalg.norm(a) < 1e-9:
                continue
            cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
            assert cos > 0.99 and abs(np.linalg.norm(b) / np.linalg.norm(a) - 1) < 0.15, (k, cos, no_cat)
        m.close()
    monkeypatch.delenv("ADN_NO_CAT", raising=False)


def test_zoo_factories_of_the_last_timestep_family(model_cls):
    """modelzoo/adenet_v3.py:64-188, deltanet.py:59-77, adenet_v2_1.py:58-175, adenet_3stream_dropout.py:13-139:
    positional signatures, layer names, widths, head."""
    from ip_avsr_amd.modelzoo import adenet_v3, deltanet, adenet_v2_1, adenet_3stream_dropout

    class Layer(object):
        def __init__(self, W, b):
            self.W, self.b = W, b

    class Net(object):                                  # what nolearn hands over: .get_all_layers()[1..4].W / .b
        def __init__(self, d, rng):
            dims = [d, 20, 12, 8, 5]
            self.layers = [None] + [Layer(rng.normal(0, 0.1, (a, b)).astype(np.float32), np.zeros(b, np.float32))
                                    for a, b in zip(dims[:-1], dims[1:])]

        def get_all_layers(self):
            return self.layers

    rng = np.random.RandomState(0)
    B, T = 4, 6
    mask = np.ones((B, T), np.uint8); mask[2, 3:] = 0
    x = lambda d: (rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32)
    net, fuse = adenet_v3.create_model(Net(30, rng), Net(30, rng), (None, None, 30), None, (None, None), None,
                                       (None, None, 9), None, (None, None, 30), None, lstm_size=4, win=None,
                                       output_classes=5, fusiontype='concat')
    names = [p.name for p in net.params]
    assert names[0] == "fc1_raw.W" and "lstm_dct.W_in_to_ingate" in names and names[-2:] == ["output.W", "output.b"]
    assert net.H == 8 and net.head == "last"                                   # lstm_size / (1 - 0.5)
    assert net.spec["streams"][1]["dropout"] == 0.2 and net.spec["agg_dropout"] == 0.5
    assert net.spec["streams"][0]["enc_acts"] == ["sigmoid", "sigmoid", "sigmoid", "linear"]
    probs = net.predict([x(30), x(9), x(30)], mask, 2)
    assert probs.shape == (B, 5) and np.allclose(probs.sum(1), 1, atol=1e-5)
    train, train_cost, test_cost, val_fn = net.compile(0.01, updates="nesterov_momentum", momentum=0.9)
    y = np.repeat(rng.randint(0, 5, size=(B, 1)), T, axis=1).astype(np.int32)
    ins = [x(30), x(9), x(30)]
    c0 = test_cost(*ins, y, mask, 2)
    for _ in range(25):
        train(*ins, y, mask, 2)
    assert test_cost(*ins, y, mask, 2) < c0                                     # it learns
    assert train_cost(*ins, y, mask, 2) != train_cost(*ins, y, mask, 2)         # dropout active in compute_train_cost
    net.close()
    d = deltanet.create_model(Net(30, rng), (None, None, 30), None, (None, None), None, lstm_size=6, win=None,
                              output_classes=3)
    assert d.head == "last" and [p.name for p in d.params][-2:] == ["output.W", "output.b"]
    assert d.predict([x(30)], mask, 2).shape == (B, 3)
    d.close()
    v, _ = adenet_v2_1.create_model(Net(30, rng), Net(30, rng), (None, None, 30), None, (None, None), None,
                                    (None, None, 30), None, lstm_size=5, output_classes=4)
    assert v.head == "last" and v.predict([x(30), x(30)], mask, 2).shape == (B, 4)
    v.close()
    ae = lambda: ([l.W for l in Net(30, rng).layers[1:]], [l.b for l in Net(30, rng).layers[1:]], [20, 12, 8, 5],
                  ["rectify", "rectify", "rectify", "linear"])
    dr, _ = adenet_3stream_dropout.create_model(ae(), ae(), ae(), (None, None, 30), None, (None, None, 30), None,
                                                (None, None, 30), None, (None, None), None, lstm_size=3, output_classes=4)
    assert dr.H == 6 and dr.head == "frames" and dr.predict([x(30)] * 3, mask, 2).shape == (B, T, 4)
    dr.close()


def test_more_zoo_factories_of_the_last_timestep_family(model_cls):
    """modelzoo/adenet_v4.py:48-147 (ONE forward aggregation LSTM), adenet_v5.py:64-186 / adenet_v6.py:64-177 (sum or
    adaptive sum, summed BLSTM), lstm_classifier_baseline.py:56-82, baseline_end2end.py:64-116 (no deltas): positional
    signatures, layer names, widths, head; the forward pass against the oracle for the single-LSTM aggregation."""
    from ip_avsr_amd.modelzoo import adenet_v4, adenet_v5, adenet_v6, lstm_classifier_baseline, baseline_end2end

    class Layer(object):
        def __init__(self, W, b):
            self.W, self.b = W, b

    class Net(object):
        def __init__(self, d, rng):
            dims = [d, 20, 12, 8, 5]
            self.layers = [None] + [Layer(rng.normal(0, 0.1, (a, b)).astype(np.float32), np.zeros(b, np.float32))
                                    for a, b in zip(dims[:-1], dims[1:])]

        def get_all_layers(self):
            return self.layers

    rng = np.random.RandomState(3)
    B, T = 5, 7
    mask = np.ones((B, T), np.uint8); mask[1, 4:] = 0; mask[3, 1:] = 0
    x = lambda d: (rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32)
    y = np.repeat(rng.randint(0, 4, size=(B, 1)), T, axis=1).astype(np.int32)

    v4, fuse = adenet_v4.create_model(Net(30, rng), (None, None, 30), None, (None, None), None, (None, None, 9), None,
                                      lstm_size=3, win=None, output_classes=4)
    names = [p.name for p in v4.params]
    assert names[0] == "fc1.W" and "lstm_bn.W_in_to_ingate" in names and "lstm_dct.W_hid_to_cell" in names
    assert "lstm_agg.W_in_to_ingate" in names and not any(n.startswith(("f_lstm_agg", "b_lstm_agg")) for n in names)
    assert v4.H == 6 and v4.head == "last" and v4.spec["fusion"] == "sum" and v4.spec["agg_dropout"] == 0.5
    ins = [x(30), x(9)]
    probs = v4.predict(ins, mask, 2)
    assert probs.shape == (B, 4)
    p64 = {k: v.astype(np.float64) for k, v in v4.get_params_dict().items()}
    ref = O.forward(v4.spec, p64, [a.astype(np.float64) for a in ins], mask, 2)
    assert np.abs(probs - ref).max() <= 2e-5
    l_ref, g_ref, _ = O.loss_and_grads(v4.spec, p64, [a.astype(np.float64) for a in ins], y, mask, 2)
    l = v4.compute_grads(ins, y, mask, 2, deterministic=True)
    assert abs(l - l_ref) <= 1e-5 * abs(l_ref)
    check_grads(v4.get_grads_dict(), g_ref, v4.spec)
    v4.close()

    for use_adascale in (False, True):
        v5, fuse = adenet_v5.create_model(Net(30, rng), Net(30, rng), (None, None, 30), None, (None, None), None,
                                          (None, None, 9), None, (None, None, 30), None, 3, None, 4, use_adascale)
        names = [p.name for p in v5.params]
        assert v5.H == 6 and v5.head == "last" and v5.spec["fusion"] == ("adasum" if use_adascale else "sum")
        assert ("adasum1.adacoeff0" in names) == use_adascale and "f_lstm_agg.W_in_to_ingate" in names
        assert v5.predict([x(30), x(9), x(30)], mask, 2).shape == (B, 4)
        v5.close()
    v6, _ = adenet_v6.create_model(Net(30, rng), Net(30, rng), (None, None, 30), None, (None, None), None,
                                   (None, None, 30), None, 3, None, 4)
    assert v6.S == 2 and v6.H == 6 and [p.name for p in v6.params][0] == "fc1_raw.W"
    assert v6.predict([x(30), x(30)], mask, 2).shape == (B, 4)
    v6.close()
    lb = lstm_classifier_baseline.create_model((None, None, 9), None, (None, None), None, lstm_size=5, output_classes=3)
    names = [p.name for p in lb.params]
    assert names[0].startswith("f_lstm.") and any(n.startswith("b_lstm.") for n in names) and names[-1] == "output.b"
    assert lb.head == "last" and lb.predict([x(9)], mask, 2).shape == (B, 3)
    lb.close()
    e2e = baseline_end2end.create_model(Net(30, rng), (None, None, 30), None, (None, None), None, lstm_size=5, output_classes=3)
    assert e2e.spec["streams"][0]["delta"] is False and "f_lstm1.W_in_to_ingate" in [p.name for p in e2e.params]
    w_in = e2e.get_param("f_lstm1.W_in_to_ingate")
    assert w_in.shape == (5, 5)                                             # the LSTM reads the 5 bottleneck features, no deltas
    assert e2e.predict([x(30)], mask, 2).shape == (B, 3)
    e2e.close()
