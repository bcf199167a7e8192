"""adenet_v1 / adenet_v1_1 on the GPU (reference modelzoo/adenet_v1.py:48-109, adenet_v1_1.py:48-114): BatchNormLayer
behind the encoder, the DCT input concatenated behind the delta features, a stream BLSTM under a (wider, in v1)
aggregation BLSTM, last-timestep head.  Against the fp64 oracle, whose BatchNorm / concat / two-width arithmetic is
pinned by finite differences and torch's batch_norm (tests/test_oracle.py)."""
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
    p["batchnorm1.inv_std"] = (np.abs(p["batchnorm1.inv_std"]) + 0.5).astype(np.float32)
    lens = rng.integers(1, T + 1, size=B); lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    s = spec["streams"][0]
    inputs = [(rng.normal(size=(B, T, s["input_dim"])) * mask[..., None]).astype(np.float32),
              (rng.normal(size=(B, T, s["aux_dim"])) * mask[..., None]).astype(np.float32)]
    y = np.repeat(rng.integers(0, spec["classes"], size=(B, 1)), T, axis=1).astype(np.int32)
    return p, inputs, y, mask


def check_grads(g, g_ref, spec, tol=1e-4):
    gscale = max(np.abs(v).max() for v in g_ref.values())
    for k in O.param_names(spec):
        err = np.abs(g[k] - g_ref[k]).max()
        # (+ 1e-6 of the gradient scale: the bias ahead of a BatchNorm layer in batch-statistics mode has an EXACT zero
        #  gradient -- the layer subtracts the batch mean -- which fp32 reproduces as cancellation noise)
        assert err <= tol * max(np.abs(g_ref[k]).max(), 1e-3 * gscale) + 1e-6 * gscale, (k, err, np.abs(g_ref[k]).max())


@pytest.mark.parametrize("v1_1", [False, True])
def test_forward_loss_gradients_and_running_averages_match_the_oracle(model_cls, v1_1):
    spec = O.spec_adenet_v1(11, 6, enc_shapes=(9, 4), enc_acts=("sigmoid", "linear"), lstm_size=3, classes=4, v1_1=v1_1)
    B, T, theta = 6, 8, 2
    p, inputs, y, mask = case(spec, B, T, seed=17 + v1_1)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    in64 = [x.astype(np.float64) for x in inputs]
    m = model_cls(spec)
    assert [q.name for q in m.params] == O.param_names(spec)
    shapes = O.param_shapes(spec)
    assert all(q.shape == tuple(shapes[q.name]) for q in m.params)          # the narrow views of v1's stream BLSTM included
    m.set_params_dict(p)
    for k in p:
        np.testing.assert_array_equal(m.get_param(k), p[k])
    # deterministic: running averages
    probs = m.predict(inputs, mask, theta)
    ref = O.forward(spec, p64, in64, mask, theta)
    assert probs.shape == (B, 4) and np.abs(probs - ref).max() <= 2e-5
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, in64, y, mask, theta, training=False)
    l = m.compute_grads(inputs, y, mask, theta, deterministic=True)
    assert abs(l - l_ref) <= 1e-5 * abs(l_ref)
    check_grads(m.get_grads_dict(), g_ref, spec)
    np.testing.assert_array_equal(m.get_param("batchnorm1.mean"), p["batchnorm1.mean"])     # untouched by a deterministic pass
    # training pass: batch statistics, their adjoint, the running-average update; dropout (v1_1) with the shared hash masks
    dr = dict(seed=4242, counter=3) if v1_1 else None
    m.set_dropout_state(4242, 3)
    l_ref, g_ref, cache = O.loss_and_grads(spec, p64, in64, y, mask, theta, dropout=dr, training=True)
    l = m.compute_grads(inputs, y, mask, theta)
    assert abs(l - l_ref) <= 1e-5 * abs(l_ref)
    check_grads(m.get_grads_dict(), g_ref, spec)
    q64 = {k: v.copy() for k, v in p64.items()}
    O.bn_running_update(spec, q64, cache)
    assert np.abs(m.get_param("batchnorm1.mean") - q64["batchnorm1.mean"]).max() <= 1e-6
    assert np.abs(m.get_param("batchnorm1.inv_std") - q64["batchnorm1.inv_std"]).max() <= 1e-5
    assert not m.get_grads_dict()["batchnorm1.mean"].any() and not m.get_grads_dict()["batchnorm1.inv_std"].any()
    names_trainable = [q.name for q in m.get_all_params(trainable=True)]
    assert "batchnorm1.gamma" in names_trainable and "batchnorm1.mean" not in names_trainable
    m.close()


def test_narrow_stream_lstm_padding_never_trains(model_cls):
    """adenet_v1: the 3-unit stream BLSTM lives in 6-unit kernels.  After Adam steps the flat parameter buffer holds
    exact zeros wherever the views do not reach (checked through the wider model's own accounting: the sum of squares
    of the whole buffer equals the sum over the exposed views), and the trajectory still follows the oracle."""
    from ip_avsr_amd.parallel import wrap_flat_buffer
    from ip_avsr_amd import _lib
    spec = O.spec_adenet_v1(11, 6, enc_shapes=(9, 4), enc_acts=("sigmoid", "linear"), lstm_size=3, classes=4)
    B, T, theta = 5, 7, 2
    p, inputs, y, mask = case(spec, B, T, seed=3)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    in64 = [x.astype(np.float64) for x in inputs]
    m = model_cls(spec)
    m.set_params_dict(p)
    st = O.adam_init(p64)
    for _ in range(4):
        l = m.train_step(inputs, y, mask, theta, 1e-2)
        l_ref = O.train_step(spec, p64, st, in64, y, mask, theta, 1e-2, training=True)
        assert abs(l - l_ref) <= 2e-5 * abs(l_ref)
    got = m.get_params_dict()
    # (the bias ahead of the BatchNorm layer is left out: its true gradient is exactly zero, fp32 leaves ~1e-8 of
    #  cancellation noise, and Adam's g / sqrt(v) turns noise of ANY size into steps of order lr -- in the reference's
    #  float32 graph as well; BatchNorm subtracts whatever it drifts to)
    skip = spec["streams"][0]["enc_names"][-1] + ".b"
    assert max(np.abs(got[k] - p64[k]).max() for k in p64 if k != skip) <= 2e-4
    flat = wrap_flat_buffer(m, _lib.BUF_PARAM).cpu().numpy().astype(np.float64)
    exposed = sum(float((v.astype(np.float64) ** 2).sum()) for v in got.values())
    assert abs(float((flat ** 2).sum()) - exposed) <= 1e-9 * exposed
    m.close()


def test_zoo_factories_adenet_v1_and_v1_1(model_cls):
    from ip_avsr_amd.modelzoo import adenet_v1, adenet_v1_1

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

    rng = np.random.RandomState(0)
    B, T = 4, 6
    mask = np.ones((B, T), np.uint8); mask[2, 3:] = 0
    x = lambda d: (rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32)
    net, concat = adenet_v1.create_model(Net(30, rng), (None, None, 30), None, (None, None), None, (None, None, 9), None,
                                         lstm_size=4, win=None, output_classes=3)
    names = [q.name for q in net.params]
    i = names.index("bottleneck.b")
    assert names[0] == "fc1.W" and names[i + 1:i + 5] == ["batchnorm1.beta", "batchnorm1.gamma", "batchnorm1.mean", "batchnorm1.inv_std"]
    assert net.get_param("f_lstm1.W_in_to_ingate").shape == (3 * 5 + 9, 4) and net.get_param("f_lstm2.W_hid_to_cell").shape == (8, 8)
    assert net.get_param("f_lstm2.W_in_to_ingate").shape == (4, 8) and "f_lstm1.W_cell_to_outgate" in names
    assert np.array_equal(net.get_param("batchnorm1.gamma"), np.ones(5, np.float32)) and net.head == "last"
    assert net.predict([x(30), x(9)], mask, 2).shape == (B, 3)
    train, train_cost, test_cost, val_fn = net.compile(1.0, updates="adadelta")
    y = np.repeat(rng.randint(0, 3, size=(B, 1)), T, axis=1).astype(np.int32)
    ins = [x(30), x(9)]
    first = train(*ins, y, mask, 2)
    for _ in range(40):
        last = train(*ins, y, mask, 2)
    assert last < first and np.isfinite(test_cost(*ins, y, mask, 2))     # (batch-statistics cost; the running averages lag by design)
    assert val_fn(*ins, mask, 2).shape == (B, 3)
    net.close()
    v11 = adenet_v1_1.create_model(Net(30, rng), (None, None, 30), None, (None, None), None, (None, None, 9), None,
                                   lstm_size=4, win=None, output_classes=3)
    assert v11.H == 8 and v11.spec["streams"][0]["dropout"] == 0.5 and v11.spec["agg_dropout"] == 0.5
    assert v11.get_param("f_lstm1.W_hid_to_cell").shape == (8, 8)
    assert train_cost is not None and v11.predict([x(30), x(9)], mask, 2).shape == (B, 3)
    assert v11.loss([x(30), x(9)], y, mask, 2, deterministic=False) != v11.loss([x(30), x(9)], y, mask, 2, deterministic=False)
    v11.close()
