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
        assert cos > 0.99 and abs(np.linalg.norm(b) / np.linalg.norm(a) - 1) < 0.1, (k, cos)
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
