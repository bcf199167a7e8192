"""The conv auto-encoder oracle (oracle/convae_oracle.py) against torch's CPU operators as an independent
second opinion -- conv2d with flipped filters, max_pool2d, interpolate(nearest), conv_transpose2d -- for the forward pass
and, through autograd, for every gradient incl. the tied weights; plus finite differences on a tiny image."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import convae_oracle as C


def torch_forward(p, x, hw):
    st = lambda t: C.SCALE_OUT * torch.tanh(C.SCALE_IN * t)
    flip = lambda W: torch.flip(W, dims=(2, 3))
    neg = float("-inf")
    B = x.shape[0]
    x0 = x.reshape(B, 1, *hw)
    a1 = st(F.conv2d(x0, flip(p["conv2d1.W"]), p["conv2d1.b"]))
    p2 = F.max_pool2d(a1, 2)
    a3 = st(F.conv2d(p2, flip(p["conv2d3.W"]), p["conv2d3.b"]))
    p4 = F.max_pool2d(F.pad(a3, (0, 0, 1, 1), value=neg), 2)
    a5 = st(F.conv2d(p4, flip(p["conv2d5.W"]), p["conv2d5.b"]))
    a7 = st(a5.reshape(B, -1) @ p["dense7.W"] + p["dense7.b"])
    code = a7 @ p["bottleneck.W"] + p["bottleneck.b"]
    a8 = code @ p["bottleneck.W"].T + p["dense8.b"]
    a9 = st(a8 @ p["dense7.W"].T + p["dense9.b"])
    r10 = a9.reshape(a5.shape)
    # the adjoint of conv2d(., flip(W)) is conv_transpose2d(., flip(W))
    a11 = st(F.conv_transpose2d(r10, flip(p["conv2d5.W"]), p["deconv2d11.b"]))
    a13 = st(F.conv_transpose2d(F.interpolate(a11, scale_factor=2, mode="nearest"), flip(p["conv2d3.W"]), p["deconv2d13.b"]))
    a15 = st(F.conv_transpose2d(F.interpolate(a13, scale_factor=2, mode="nearest"), flip(p["conv2d1.W"]), p["deconv2d14.b"],
                                padding=(1, 0)))
    return a15.reshape(B, -1), code


@pytest.mark.parametrize("hw", [(30, 40), (22, 28)])
def test_forward_and_gradients_match_torch(hw):
    rng = np.random.default_rng(5)
    p = C.init_params(rng, np.float64, dense=40, bottleneck=7, image_hw=hw, bias_noise=0.1)
    B = 3
    x = rng.normal(size=(B, hw[0] * hw[1]))
    loss, g, cache = C.loss_and_grads(p, x, image_hw=hw)
    assert cache["recon"].shape == x.shape and C.geometry(hw)["d15"] == hw
    pt = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    recon_t, code_t = torch_forward(pt, torch.tensor(x), hw)
    assert np.abs(recon_t.detach().numpy() - cache["recon"]).max() < 1e-10
    assert np.abs(code_t.detach().numpy() - cache["code"]).max() < 1e-10
    lt = ((recon_t - torch.tensor(x)) ** 2).mean()
    assert abs(lt.item() - loss) < 1e-12
    lt.backward()
    for k in C.param_names():
        assert np.abs(pt[k].grad.numpy() - g[k]).max() <= 1e-10 * max(1.0, np.abs(g[k]).max()), k


def test_finite_differences_and_param_order():
    hw = (22, 28)
    rng = np.random.default_rng(9)
    p = C.init_params(rng, np.float64, dense=12, bottleneck=4, image_hw=hw, bias_noise=0.1)
    assert list(p) == C.param_names()
    x = rng.normal(size=(2, hw[0] * hw[1]))
    loss, g, _ = C.loss_and_grads(p, x, image_hw=hw)
    for k in C.param_names():
        idx = tuple(rng.integers(0, n) for n in p[k].shape)
        q = {a: b.copy() for a, b in p.items()}
        q[k][idx] += 1e-6
        lp = C.loss_and_grads(q, x, image_hw=hw)[0]
        q[k][idx] -= 2e-6
        lm = C.loss_and_grads(q, x, image_hw=hw)[0]
        fd = (lp - lm) / 2e-6
        assert abs(fd - g[k][idx]) <= 1e-7 + 1e-5 * abs(fd), (k, fd, g[k][idx])


def test_padded_pooling_never_selects_padding():
    x = -np.ones((1, 1, 3, 4))                          # all negative: zero padding would win if it took part
    y, arg = C.maxpool2(x, pad=(1, 0))
    assert y.shape == (1, 1, 2, 2) and (y == -1).all()
    back = C.maxpool2_bwd(np.ones_like(y), arg, (3, 4), pad=(1, 0))
    assert back.sum() == 4 and back.shape == x.shape


# ---------------------------------------------------------------------------------------------------------------------
# the BatchNorm / dropout variants (modelzoo/avletters_convae_{bn,drop,bndrop}.py)
# ---------------------------------------------------------------------------------------------------------------------
def torch_forward_variant(p, x, hw, variant, masks, training):
    """The same graph from torch's operators: BatchNorm as F.batch_norm (biased variance, eps inside the root -- what
    Lasagne's inv_std = 1 / sqrt(var + eps) is), dropout as a multiplication with the oracle's own mask / (1 - p)."""
    v = C.VARIANTS[variant]
    n, (si, so) = v["names"], v["tanh"]
    st = lambda t: so * torch.tanh(si * t)
    flip = lambda W: torch.flip(W, dims=(2, 3))
    neg = float("-inf")
    B = x.shape[0]

    def bn(k, t):
        name = v["bn_names"][k]
        if training:
            return F.batch_norm(t, None, None, p[name + ".gamma"], p[name + ".beta"], True, 0.0, C.BN_EPS)
        shp = (1, -1) + (1,) * (t.dim() - 2)
        return (t - p[name + ".mean"].reshape(shp)) * p[name + ".inv_std"].reshape(shp) * p[name + ".gamma"].reshape(shp) + \
            p[name + ".beta"].reshape(shp)

    dr = lambda k, t: t * torch.tensor(masks[k])
    t = dr(0, x.reshape(B, 1, *hw))
    a1 = st(F.conv2d(t, flip(p[n["c1"] + ".W"]), p[n["c1"] + ".b"]))
    t = bn(0, a1) if v["bn"] == "conv" else a1
    t = F.max_pool2d(t, 2)
    t = dr(1, bn(0, t) if v["bn"] == "pool" else t)
    a3 = st(F.conv2d(t, flip(p[n["c3"] + ".W"]), p[n["c3"] + ".b"]))
    t = bn(1, a3) if v["bn"] == "conv" else a3
    t = F.max_pool2d(F.pad(t, (0, 0, 1, 1), value=neg), 2)
    t = dr(2, bn(1, t) if v["bn"] == "pool" else t)
    a5 = st(F.conv2d(t, flip(p[n["c5"] + ".W"]), p[n["c5"] + ".b"]))
    t = (bn(2, a5) if v["bn"] == "conv" else a5).reshape(B, -1)
    t = dr(3, bn(2, t) if v["bn"] == "pool" else t)
    a7 = st(t @ p[n["d7"] + ".W"] + p[n["d7"] + ".b"])
    t = dr(4, bn(3, a7) if v["bn"] else a7)
    code = t @ p["bottleneck.W"] + p["bottleneck.b"]
    a8 = code @ p["bottleneck.W"].T + p[n["d8"] + ".b"]
    a9 = st(a8 @ p[n["d7"] + ".W"].T + p[n["d9"] + ".b"])
    a11 = st(F.conv_transpose2d(a9.reshape(a5.shape), flip(p[n["c5"] + ".W"]), p[n["dc11"] + ".b"]))
    a13 = st(F.conv_transpose2d(F.interpolate(a11, scale_factor=2, mode="nearest"), flip(p[n["c3"] + ".W"]), p[n["dc13"] + ".b"]))
    a15 = st(F.conv_transpose2d(F.interpolate(a13, scale_factor=2, mode="nearest"), flip(p[n["c1"] + ".W"]), p[n["dc15"] + ".b"],
                                padding=(1, 0)))
    return a15.reshape(B, -1), code


@pytest.mark.parametrize("variant", ["batchnorm", "dropout", "bn+dropout"])
@pytest.mark.parametrize("training", [True, False])
def test_variants_match_torch(variant, training):
    hw = (22, 28)
    rng = np.random.default_rng(11)
    p = C.init_params(rng, np.float64, dense=24, bottleneck=6, image_hw=hw, bias_noise=0.1, variant=variant)
    assert list(p) == C.param_names(variant)
    B = 4
    x = rng.normal(size=(B, hw[0] * hw[1]))
    dropout = dict(seed=77, counter=3) if training else None
    loss, g, c = C.loss_and_grads(p, x, image_hw=hw, variant=variant, dropout=dropout, training=training)
    masks = [c["drop%d" % k] for k in range(5)]
    if training and C.VARIANTS[variant]["drop"]:
        kept = masks[1].astype(bool).mean()
        assert 0.4 < kept < 0.6 and set(np.unique(masks[0])) == {0.0, 1.25}
    else:
        assert all((m == 1).all() for m in masks)
    pt = {k: torch.tensor(v_, requires_grad=True) for k, v_ in p.items()}
    recon_t, code_t = torch_forward_variant(pt, torch.tensor(x), hw, variant, masks, training)
    assert np.abs(recon_t.detach().numpy() - c["recon"]).max() < 1e-9
    assert np.abs(code_t.detach().numpy() - c["code"]).max() < 1e-9
    lt = ((recon_t - torch.tensor(x)) ** 2).mean()
    assert abs(lt.item() - loss) < 1e-12
    lt.backward()
    for k in C.trainable_names(variant):
        assert np.abs(pt[k].grad.numpy() - g[k]).max() <= 1e-9 * max(1.0, np.abs(g[k]).max()), k
    for k in set(C.param_names(variant)) - set(C.trainable_names(variant)):
        assert not g[k].any()                        # running averages: not trainable


def test_batchnorm_running_averages_follow_lasagne():
    hw = (22, 28)
    rng = np.random.default_rng(2)
    p = C.init_params(rng, np.float64, dense=12, bottleneck=4, image_hw=hw, variant="batchnorm")
    x = rng.normal(size=(3, hw[0] * hw[1]))
    _, _, c = C.forward(p, x, hw, want_cache=True, variant="batchnorm", training=True)
    before = p["batchnorm8.inv_std"].copy()
    C.bn_running_update(p, c, "batchnorm")
    flat = c["a5"].reshape(3, -1)
    np.testing.assert_allclose(p["batchnorm8.mean"], 0.1 * flat.mean(0), atol=1e-12)
    np.testing.assert_allclose(p["batchnorm8.inv_std"], 0.9 * before + 0.1 / np.sqrt(flat.var(0) + 1e-4), atol=1e-12)
    assert p["batchnorm2.mean"].shape == (100,) and p["batchnorm8.mean"].shape == (C.geometry(hw)["flat"],)
