"""Conv auto-encoder on the GPU (csrc/convae.hip) against oracle/convae_oracle.py -- itself pinned to
torch's CPU convolution / pooling / transposed-convolution operators by tests/test_convae_oracle.py.  fp32 mode:
reconstruction and code within 1e-4 (the north-star bar for encoder activations), every gradient tensor within 1e-4
relative, incl. the tied weights that collect gradient from the encoder and the decoder."""
import numpy as np
import pytest

from oracle import convae_oracle as CO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ConvAE():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ip_avsr_amd.convae import ConvAE
    return ConvAE


def make(hw, dense, nb, seed, B):
    rng = np.random.default_rng(seed)
    p = CO.init_params(rng, np.float32, dense=dense, bottleneck=nb, image_hw=hw, bias_noise=0.05)
    x = rng.normal(size=(B, hw[0] * hw[1])).astype(np.float32)
    return p, x


@pytest.mark.parametrize("hw,dense,nb,B", [((30, 40), 500, 50, 5), ((22, 28), 24, 6, 3), ((26, 44), 40, 10, 2), ((34, 28), 32, 8, 1),
                                           ((38, 48), 20, 5, 3), ((22, 36), 16, 4, 2)])   # (H = 2 mod 4, W = 0 mod 4: the sizes the decoder reproduces)
def test_forward_loss_and_gradients_match_the_oracle(ConvAE, hw, dense, nb, B):
    p, x = make(hw, dense, nb, 3, B)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    m = ConvAE(hw, dense, nb)
    assert m.param_names == CO.param_names()
    m.set_params_dict(p)
    for k in p:
        np.testing.assert_array_equal(m.get_param(k), p[k])                # layout conversion round trip is exact
    recon_ref, code_ref = CO.forward(p64, x.astype(np.float64), hw)
    assert np.abs(m.recon_fn(x) - recon_ref).max() <= 1e-4
    assert np.abs(m.encode(x) - code_ref).max() <= 1e-4
    loss_ref, g_ref, _ = CO.loss_and_grads(p64, x.astype(np.float64), image_hw=hw)
    assert abs(m.cost(x) - loss_ref) <= 1e-5 * loss_ref
    loss = m.compute_grads(x)
    assert abs(loss - loss_ref) <= 1e-5 * loss_ref
    g = m.get_grads_dict()
    gscale = max(np.abs(v).max() for v in g_ref.values())
    for k in CO.param_names():
        err = np.abs(g[k] - g_ref[k]).max()
        assert err <= 1e-4 * max(np.abs(g_ref[k]).max(), 1e-3 * gscale) + 1e-10, (k, err, np.abs(g_ref[k]).max())
    # a separate target (denoising form) and device tensors
    import torch
    t = np.random.default_rng(1).normal(size=x.shape).astype(np.float32)
    l2 = CO.loss_and_grads(p64, x.astype(np.float64), t.astype(np.float64), hw)[0]
    assert abs(m.cost(torch.as_tensor(x, device="cuda"), torch.as_tensor(t, device="cuda")) - l2) <= 1e-5 * l2
    m.close()


def test_adadelta_training_reduces_the_error_and_matches_the_oracle_steps(ConvAE):
    from oracle import adenet_oracle as O
    hw = (22, 28)
    p, x = make(hw, 20, 5, 11, 6)
    m = ConvAE(hw, 20, 5)
    m.set_params_dict(p)
    ref = {k: v.copy() for k, v in p.items()}
    st = O.adadelta_init(ref)
    first = None
    for step in range(3):
        loss_ref, g, _ = CO.loss_and_grads(ref, x, image_hw=hw)
        g = {k: np.asarray(v, np.float32) for k, v in g.items()}
        O.adadelta_step(ref, g, st, 0.8, 0.95, 1e-6)
        loss = m.train(x, learning_rate=0.8)
        first = first if first is not None else loss
        assert abs(loss - loss_ref) <= 1e-4 * abs(loss_ref)
    for k in ref:
        assert np.abs(m.get_param(k) - ref[k]).max() <= 5e-4 * max(np.abs(ref[k]).max(), 1e-2), k
    m.close()
    # it learns: smooth images in [-1, 1], Adam on the same gradients
    yy, xx = np.mgrid[0:hw[0], 0:hw[1]]
    imgs = np.stack([np.sin(0.2 * (k + 1) * xx / 3 + k) * np.cos(0.15 * (k + 2) * yy / 2) for k in range(8)]).reshape(8, -1)
    m = ConvAE(hw, 20, 5)
    m.init_params(np.random.RandomState(0))
    start = m.cost(imgs.astype(np.float32))
    for _ in range(80):
        m.compute_grads(imgs.astype(np.float32))
        m.apply_adam(2e-3)
    assert m.cost(imgs.astype(np.float32)) < 0.6 * start
    m.close()


def test_zoo_factory_and_bf16_mode(ConvAE):
    from ip_avsr_amd.modelzoo import avletters_convae
    np.random.seed(12)                               # (the factory draws GlorotUniform weights from the global generator, as Lasagne does)
    net, enc = avletters_convae.create_model((None, 1, 30, 40), {"BOTTLENECK": 50, "DENSE": 500})
    x = np.random.default_rng(0).normal(size=(4, 1200)).astype(np.float32)
    assert net.recon_fn(x).shape == (4, 1200) and enc(x).shape == (4, 50)
    vals = net.get_all_param_values()
    assert [v.shape for v in vals[:2]] == [(100, 1, 5, 5), (100,)] and len(vals) == 15
    b16, _ = avletters_convae.create_model((None, 1, 30, 40), {"BOTTLENECK": 50, "DENSE": 500, "PRECISION": "bf16"})
    b16.set_all_param_values(vals)
    r32, r16 = net.recon_fn(x), b16.recon_fn(x)
    assert np.abs(r16 - r32).max() <= 5e-2 * max(1.0, np.abs(r32).max())
    l32, l16 = net.compute_grads(x), b16.compute_grads(x)                # heavy layers on the bf16-operand GEMM kernels
    assert abs(l16 - l32) <= 2e-2 * l32
    g32, g16 = net.get_grads_dict(), b16.get_grads_dict()
    for k in g32:
        a, b = g32[k].ravel().astype(np.float64), g16[k].ravel().astype(np.float64)
        cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
        assert cos > 0.99 and abs(np.linalg.norm(b) / np.linalg.norm(a) - 1) < 0.1, (k, cos, np.linalg.norm(b) / np.linalg.norm(a))
    net.close(); b16.close()
    with pytest.raises(Exception):
        ConvAE((10, 10), 8, 2)


def test_image_sizes_the_decoder_cannot_reproduce_are_rejected(ConvAE):
    """conv / pool / upscale / deconv only give back H x W for H = 2 (mod 4), W = 0 (mod 4); anything else must fail at
    construction, not train on a mismatched reconstruction (the reference would fail on the shape of its MSE)."""
    from ip_avsr_amd._lib import AdenetError
    for hw in ((30, 50), (29, 33), (32, 40)):
        with pytest.raises(AdenetError):
            ConvAE(hw, 16, 4)


def test_trainer_loop_of_the_reference_script(ConvAE, tmp_path):
    """avletters/avletters_convae.py:202-329 on synthetic 22 x 28 'mouth' images: the loop runs, the reconstruction error
    falls, the learning rate decays by 0.9 after every epoch past the 11th, both parameter lists are saved."""
    from ip_avsr_amd.avletters import avletters_convae as T
    from ip_avsr_amd.utils.io import load_model
    rng = np.random.RandomState(0)
    yy, xx = np.mgrid[0:22, 0:28]

    def blobs(n):
        cx, cy, s = rng.uniform(8, 20, n), rng.uniform(6, 16, n), rng.uniform(2, 5, n)
        im = np.exp(-((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) / (2 * s[:, None, None] ** 2))
        im = im.reshape(n, -1)
        return ((im - im.mean(1, keepdims=True)) / im.std(1, keepdims=True)).reshape(n, 22, 28).astype(np.float32)

    X, X_val = blobs(600), blobs(150)
    prefix = str(tmp_path / "conv")
    out = T.main(["--epochs", "14", "--epoch_size", "6", "--dense", "48", "--bottleneck", "12", "--image", "44,56,22,28",
                  "--save_prefix", prefix, "--seed", "1"], data=(X, X_val))
    assert len(out["costs"]) == 14 and np.isfinite(out["costs"]).all() and np.isfinite(out["val_costs"]).all()
    assert out["val_costs"][-1] < 0.9 * out["val_costs"][0]
    assert abs(out["learning_rate"] - 0.8 * 0.9 ** 3) < 1e-6                  # epochs 12, 13, 14 (index > 10) decayed it
    enc, ae = load_model(prefix + "_encoder.dat"), load_model(prefix + "_ae.dat")
    assert len(enc) == 10 and len(ae) == len(out["network"].param_names)
    np.testing.assert_array_equal(ae[0], out["network"].get_param("conv2d1.W"))
    assert out["recon"].shape == (150, 22 * 28)
    out["network"].close()


def test_conv_encoder_as_the_front_end_of_a_stream(ConvAE):
    """The bottleneck of a conv auto-encoder offered as a stream's ``ae`` (modelzoo/_factory.stream): frames -> 12-d code on
    the GPU -> delta layer -> LSTM.  Equal to feeding the codes to an encoder-less stream by hand, for host and device
    inputs; the second stream keeps its dense encoder."""
    import torch
    from ip_avsr_amd.modelzoo import _factory as F, avletters_convae
    hw = (22, 28)
    net, bottleneck = avletters_convae.create_model((None, 1) + hw, {"DENSE": 32, "BOTTLENECK": 12})
    rng = np.random.RandomState(5)
    dims = [40, 16, 8]
    dense = ([rng.normal(0, 0.3, (a, b)).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])],
             [np.zeros(b, np.float32) for b in dims[1:]], dims[1:], ["rectify", "linear"])
    D = hw[0] * hw[1]
    las_init = __import__("ip_avsr_amd.init", fromlist=["x"])
    las_init.set_rng(np.random.RandomState(11))
    streams = [F.stream((None, None, D), bottleneck, "_conv", lstm_names=["lstm_conv"]),
               F.stream((None, None, 40), dense, "_dct", lstm_names=["lstm_dct"])]
    model, _ = F.build(streams, 10, 4, "sum", {"sum": "sum1"}, ["f_lstm_agg", "b_lstm_agg"], False, "glorot")
    las_init.set_rng(np.random.RandomState(11))
    plain = [F.stream((None, None, 12), None, "_conv", lstm_names=["lstm_conv"]),
             F.stream((None, None, 40), dense, "_dct", lstm_names=["lstm_dct"])]
    ref, _ = F.build(plain, 10, 4, "sum", {"sum": "sum1"}, ["f_lstm_agg", "b_lstm_agg"], False, "glorot")
    ref.set_all_param_values(model.get_all_param_values())
    B, T = 4, 7
    mask = np.ones((B, T), np.uint8); mask[1, 4:] = 0
    frames = (rng.normal(size=(B, T, D)) * mask[..., None]).astype(np.float32)
    other = (rng.normal(size=(B, T, 40)) * mask[..., None]).astype(np.float32)
    codes = net.encode(frames.reshape(B * T, D)).reshape(B, T, 12)
    want = ref.predict([codes, other], mask, 2)
    got = model.predict([frames, other], mask, 2)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    got_dev = model.predict([torch.tensor(frames, device="cuda"), torch.tensor(other, device="cuda")], mask, 2)
    np.testing.assert_allclose(got_dev, want, rtol=0, atol=1e-6)
    y = np.repeat(rng.randint(0, 4, size=(B, 1)), T, axis=1).astype(np.int32)
    l0 = model.loss([frames, other], y, mask, 2)
    for _ in range(20):
        model.train_step([frames, other], y, mask, 2, 1e-2)
    assert model.loss([frames, other], y, mask, 2) < l0                        # the part behind the frozen encoder learns
    with pytest.raises(ValueError):
        F.stream((None, None, 99), bottleneck)                                 # frame size mismatch
    model.close(); ref.close(); net.close()


def test_conv_front_end_in_the_four_stream_512_unit_model_against_the_oracle(ConvAE):
    """BASELINE configs[4] as one graph: runners/4stream.py's adenet_4stream (concat fusion, 512-unit stream LSTMs and BLSTM)
    with the mouth-ROI stream entering through the (frozen) conv auto-encoder's bottleneck instead of a dense encoder.
    Forward probabilities, loss and every gradient against the fp64 oracle of the SAME 4-stream graph fed the oracle-side
    conv codes (oracle/convae_oracle.py -> oracle/adenet_oracle.py): f32 and bf16x3 arithmetic inside the 1e-4 gate, identical
    votes; bf16 tracks."""
    from ip_avsr_amd.modelzoo import adenet_4stream, avletters_convae
    from oracle import adenet_oracle as O
    hw = (22, 28)
    D = hw[0] * hw[1]
    rng = np.random.default_rng(77)
    pc = CO.init_params(rng, np.float32, dense=32, bottleneck=12, image_hw=hw, bias_noise=0.05)
    net, bottleneck = avletters_convae.create_model((None, 1) + hw, {"DENSE": 32, "BOTTLENECK": 12})
    net.set_params_dict(pc)

    def dense(d_in):
        dims = [d_in, 48, 24, 10]
        return ([rng.normal(0, 0.2, (a, b)).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])],
                [rng.normal(0, 0.05, b).astype(np.float32) for b in dims[1:]], dims[1:], ["rectify", "rectify", "linear"])
    dims = [D, 36, 44, 30]
    aes = [bottleneck, dense(36), dense(44), dense(30)]
    shapes = [(None, None, d) for d in dims]
    model, _ = adenet_4stream.create_model(aes[0], aes[1], aes[2], aes[3], shapes[0], None, shapes[1], None, shapes[2], None,
                                           shapes[3], None, (None, None), None, 512, None, 10, "concat", w_init_fn="glorot",
                                           use_peepholes=True)
    # the same graph for the oracle: stream 1 is an encoder-less 12-d stream fed the conv codes
    spec = O.spec_nstream([12, 36, 44, 30], enc_shapes=(48, 24, 10), enc_acts=("rectify", "rectify", "linear"), lstm_size=512,
                          classes=10, fusion="concat", has_encoder=[False, True, True, True], peepholes=True)
    names = O.param_names(spec)
    got_names = [p.name for p in model.params]
    assert len(names) == len(got_names)
    p64 = {n: np.asarray(model.get_param(g), np.float64) for n, g in zip(names, got_names)}
    B, T, theta = 5, 8, 3
    lens = np.array([8, 5, 8, 3, 6])
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    frames = (rng.normal(size=(B, T, D)) * mask[..., None]).astype(np.float32)
    others = [(rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32) for d in dims[1:]]
    y = np.repeat(rng.integers(0, 10, size=(B, 1)), T, axis=1).astype(np.int32)
    pc64 = {k: v.astype(np.float64) for k, v in pc.items()}
    _, codes = CO.forward(pc64, frames.reshape(B * T, D).astype(np.float64), hw)
    x64 = [codes.reshape(B, T, 12)] + [x.astype(np.float64) for x in others]
    probs_ref = O.forward(spec, p64, x64, mask, theta)
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, x64, y, mask, theta)
    gscale = max(np.abs(v).max() for v in g_ref.values())
    for prec, tol_p, tol_g in (("f32", 1e-4, 2e-4), ("bf16x3", 1e-4, 2e-4), ("bf16", 3e-2, None)):
        model.set_precision(prec)
        probs = model.predict([frames] + others, mask, theta)
        assert np.abs(probs - probs_ref).max() <= tol_p, prec
        if prec != "bf16":
            np.testing.assert_array_equal(O.majority_vote(probs, mask), O.majority_vote(probs_ref, mask))
        l = model.compute_grads([frames] + others, y, mask, theta)
        assert abs(l - l_ref) <= (2e-2 if prec == "bf16" else 2e-5) * abs(l_ref), prec
        if tol_g:
            g = model.get_grads_dict()
            for n, gn in zip(names, got_names):
                assert np.abs(g[gn] - g_ref[n]).max() <= tol_g * max(np.abs(g_ref[n]).max(), 1e-3 * gscale), (prec, n)
    model.close(); net.close()


# ---------------------------------------------------------------------------------------------------------------------
# BatchNorm / dropout variants (modelzoo/avletters_convae_{bn,drop,bndrop}.py)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant,hw,dense,nb,B", [("batchnorm", (30, 40), 500, 50, 6), ("batchnorm", (22, 28), 24, 8, 5),
                                                   ("dropout", (30, 40), 1000, 100, 4), ("dropout", (22, 28), 24, 6, 3),
                                                   ("bn+dropout", (30, 40), 500, 50, 6), ("bn+dropout", (26, 44), 40, 12, 4)])
def test_variants_match_the_oracle(ConvAE, variant, hw, dense, nb, B):
    """Deterministic passes (running averages, no masks) and non-deterministic ones (batch statistics + running-average
    update, the oracle's own masks) against the fp64 oracle: reconstruction / code 1e-4, every gradient 1e-4 of scale."""
    rng = np.random.default_rng(21)
    p = CO.init_params(rng, np.float32, dense=dense, bottleneck=nb, image_hw=hw, bias_noise=0.05, variant=variant)
    x = rng.normal(size=(B, hw[0] * hw[1])).astype(np.float32)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    x64 = x.astype(np.float64)
    m = ConvAE(hw, dense, nb, variant=variant)
    assert m.param_names == CO.param_names(variant)
    m.set_params_dict(p)
    for k in p:
        np.testing.assert_array_equal(m.get_param(k), p[k])
    # deterministic: recon_fn / eval_cost_fn
    recon_ref, code_ref = CO.forward(p64, x64, hw, variant=variant)
    assert np.abs(m.recon_fn(x) - recon_ref).max() <= 1e-4
    assert np.abs(m.encode(x) - code_ref).max() <= 1e-4
    loss_ref, g_ref, _ = CO.loss_and_grads(p64, x64, image_hw=hw, variant=variant, training=False)
    assert abs(m.cost(x) - loss_ref) <= 1e-5 * loss_ref
    assert abs(m.compute_grads(x, deterministic=True) - loss_ref) <= 1e-5 * loss_ref
    _check_grads(m, g_ref, variant)
    for k in p:                                       # a deterministic pass leaves the running averages alone
        np.testing.assert_array_equal(m.get_param(k), p[k])
    # non-deterministic: train / train_cost_fn
    drop = dict(seed=4242, counter=7)
    m.set_dropout_state(drop["seed"], drop["counter"])
    loss_ref, g_ref, c = CO.loss_and_grads(p64, x64, image_hw=hw, variant=variant, dropout=drop, training=True)
    loss = m.compute_grads(x)
    assert abs(loss - loss_ref) <= 2e-5 * loss_ref, (loss, loss_ref)
    _check_grads(m, g_ref, variant)
    CO.bn_running_update(p64, c, variant)             # ... and it moved the running averages like Lasagne's default updates
    for k in p:
        assert np.abs(m.get_param(k) - p64[k]).max() <= 1e-5 * max(1.0, np.abs(p64[k]).max()), k
    if CO.VARIANTS[variant]["drop"]:                  # the counter advanced: the next stochastic cost draws other masks
        l_next = m.cost(x, deterministic=False)
        ref_next = CO.loss_and_grads(p64, x64, image_hw=hw, variant=variant, dropout=dict(seed=4242, counter=8), training=True)[0]
        assert abs(l_next - ref_next) <= 2e-5 * ref_next and abs(l_next - loss) > 1e-6 * loss
    m.close()


def _check_grads(m, g_ref, variant):
    g = m.get_grads_dict()
    gscale = max(np.abs(v).max() for v in g_ref.values())
    for k in CO.param_names(variant):
        err = np.abs(g[k] - g_ref[k]).max()
        # (+ 3e-7 of the largest gradient: a bias in front of a BatchNorm has an almost-zero gradient -- a difference of
        #  large sums -- whose fp32 float-atomic statistics carry ~1e-7 of arrival-order noise; 1 run in 12 missed 1e-4 of
        #  1e-3 gscale by 6 %)
        assert err <= 1e-4 * max(np.abs(g_ref[k]).max(), 1e-3 * gscale) + 3e-7 * gscale, (k, err, np.abs(g_ref[k]).max())


def test_variant_zoo_factories_and_trainer(ConvAE, tmp_path):
    from ip_avsr_amd.modelzoo import avletters_convae_bn, avletters_convae_bndrop, avletters_convae_drop
    x = np.random.default_rng(0).normal(size=(4, 1200)).astype(np.float32)
    opts = {"BOTTLENECK": 50, "DENSE": 500}
    net, enc = avletters_convae_bn.create_model((None, 1, 30, 40), opts)
    assert "batchnorm8.inv_std" in net.param_names and net.param_shapes["batchnorm8.mean"] == (3000,)
    assert net.param_names[:6] == ["conv2d1.W", "conv2d1.b", "batchnorm2.beta", "batchnorm2.gamma", "batchnorm2.mean", "batchnorm2.inv_std"]
    assert enc(x).shape == (4, 50) and len(enc.get_all_param_values()) == 10 + 16
    net, enc = avletters_convae_drop.create_model((None, 1, 30, 40), opts)
    assert net.param_shapes["conv2d3.W"] == (300, 125, 5, 5) and enc(x).shape == (4, 100)     # widths / keep probability
    assert net.param_shapes["dense7.W"] == (400 * 3 * 5, 1000)
    net, enc = avletters_convae_bndrop.create_model((None, 1, 30, 40), opts)
    assert net.param_names[2] == "batchnorm1.beta" and net.recon_fn(x).shape == (4, 1200)
    # the trainer with --model bn+dropout on a few synthetic frames: runs, saves, the training cost is stochastic
    from ip_avsr_amd.avletters import avletters_convae as trainer
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, size=(40, 1200)).astype(np.float32)
    prefix = str(tmp_path / "cae")
    out = trainer.main(["--model", "bn+dropout", "--epochs", "1", "--epoch_size", "2", "--save_prefix", prefix, "--seed", "5",
                        "--bottleneck", "8", "--dense", "32"], data=(X[:32], X[32:]))
    import os
    assert os.path.exists(prefix + "_ae.dat") and os.path.exists(prefix + "_encoder.dat")
    assert np.isfinite(out["costs"]).all() and np.isfinite(out["val_costs"]).all()
    out["network"].close()


@pytest.mark.parametrize("variant", ["normal", "batchnorm", "dropout", "bn+dropout"])
@pytest.mark.parametrize("stochastic", [False, True])
def test_bf16_gradients_follow_the_f32_ones_in_every_variant(ConvAE, variant, stochastic):
    """bf16 mode (round 4: the pooling's adjoint from the pooled grid with act' from the pooled values, fused bias sums and the
    gradient's bf16 copy; the input gradients' patch matrices as bf16 only) against f32 mode from the same parameters, masks and
    batch statistics: every gradient tensor points the same way (cosine) with the same length, also with dropout rescaling the
    pooled tensors in place (the path that must NOT read act' from them)."""
    rng = np.random.default_rng(5)
    hw, dense, nb, B = (30, 40), 120, 20, 6
    p = CO.init_params(rng, np.float32, dense=dense, bottleneck=nb, image_hw=hw, bias_noise=0.05, variant=variant)
    x = np.tanh(rng.normal(size=(B, hw[0] * hw[1]))).astype(np.float32)
    out = {}
    for prec in ("f32", "bf16"):
        m = ConvAE(hw, dense, nb, precision=prec, variant=variant)
        m.set_params_dict(p)
        m.set_dropout_state(77, 3)
        loss = m.compute_grads(x, deterministic=not stochastic)
        out[prec] = (float(loss), m.get_grads_dict())
        m.close()
    assert abs(out["bf16"][0] - out["f32"][0]) <= 3e-2 * out["f32"][0]
    for k, g in out["f32"][1].items():
        a, b = g.ravel().astype(np.float64), out["bf16"][1][k].ravel().astype(np.float64)
        if np.linalg.norm(a) < 1e-12:
            assert np.linalg.norm(b) < 1e-9, k
            continue
        cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
        assert cos > 0.985 and abs(np.linalg.norm(b) / np.linalg.norm(a) - 1) < 0.1, (variant, k, cos, np.linalg.norm(b) / np.linalg.norm(a))
