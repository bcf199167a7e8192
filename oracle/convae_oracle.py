"""CPU oracle for the convolutional auto-encoder.  TEST INFRASTRUCTURE ONLY: only tests/,
__graft_entry__.smoke() and bench legs that time a CPU baseline may import it.

NumPy restatement of reference modelzoo/avletters_convae.py:33-69 (the 'normal' model) and of its training step
(avletters/avletters_convae.py:254-262: mean squared error of the reconstruction, lasagne.updates.adadelta):

    (B,1,30,40) -> conv 5x5 (100) -> maxpool 2 -> conv 5x5 (150) -> maxpool 2 pad (1,0) -> conv 3x3 (200) -> 3000
                -> dense 500 -> bottleneck 50 (linear)
                -> dense8 (W = bottleneck.W^T, linear) -> dense9 (W = dense7.W^T) -> (200,3,5)
                -> deconv 3x3 (tied conv5.W) -> upscale 2 -> deconv 5x5 (tied conv3.W) -> upscale 2
                -> deconv 5x5 crop (1,0) (tied conv1.W) -> 1200
    every nonlinearity ScaledTanh(0.5, 2.4) = 2.4 tanh(0.5 x) except the two marked linear.

PARITY STATUS: parity unpinned at the Theano/Lasagne boundary (same reason as adenet_oracle.py).  What pins it:
torch's CPU conv2d / conv_transpose2d / max_pool2d / interpolate as an independent second opinion for every layer and
for the gradients (tests/test_convae_oracle.py), plus finite differences.  [upstream] semantics restated:
  * Conv2DLayer: valid, stride 1, flip_filters=True (true convolution), W (out, in, kh, kw), b per filter;
  * MaxPool2DLayer(pool 2, ignore_border=True, pad): padded cells never win (Theano's C implementation skips them);
  * Deconv2DLayer(W=conv.W, flip_filters=not conv.flip_filters, crop): the exact adjoint of that convolution (with
    `crop` = the convolution's pad), plus its own bias;
  * Upscale2DLayer: every pixel repeated 2x2;
  * ReshapeLayer([0], -1) flattens (C,H,W) in that order.
Arrays are NCHW here, like the reference.
"""
from __future__ import annotations

import numpy as np

SCALE_IN, SCALE_OUT = 0.5, 2.4            # modelzoo/avletters_convae.py:7-26
FILTERS = (100, 150, 200)
KSIZE = (5, 5, 3)


def stanh(x):
    return x.dtype.type(SCALE_OUT) * np.tanh(x.dtype.type(SCALE_IN) * x)


def stanh_grad_from_output(y):
    """d/dx [2.4 tanh(0.5 x)] = 1.2 (1 - (y / 2.4)^2)."""
    t = y / y.dtype.type(SCALE_OUT)
    return y.dtype.type(SCALE_IN * SCALE_OUT) * (1 - t * t)


# --------------------------------------------------------------------------- layers
def im2col(x, kh, kw):
    """x (B,C,H,W) -> (B, OH, OW, C, kh, kw) view-copy of every valid patch."""
    B, C, H, W = x.shape
    OH, OW = H - kh + 1, W - kw + 1
    s = x.strides
    v = np.lib.stride_tricks.as_strided(x, (B, OH, OW, C, kh, kw), (s[0], s[2], s[3], s[1], s[2], s[3]))
    return np.ascontiguousarray(v)


def conv_valid(x, W, b):
    """True convolution (filters flipped), valid, stride 1: [upstream] Conv2DLayer defaults."""
    O, C, kh, kw = W.shape
    cols = im2col(x, kh, kw).reshape(-1, C * kh * kw)
    Wf = W[:, :, ::-1, ::-1].reshape(O, -1)
    out = cols @ Wf.T + b
    B, _, H, Wd = x.shape
    return out.reshape(B, H - kh + 1, Wd - kw + 1, O).transpose(0, 3, 1, 2)


def conv_valid_bwd(x, W, dy):
    """-> (dx, dW, db) of conv_valid."""
    O, C, kh, kw = W.shape
    B, _, H, Wd = x.shape
    OH, OW = H - kh + 1, Wd - kw + 1
    dyr = dy.transpose(0, 2, 3, 1).reshape(-1, O)
    cols = im2col(x, kh, kw).reshape(-1, C * kh * kw)
    dWf = dyr.T @ cols
    dW = dWf.reshape(O, C, kh, kw)[:, :, ::-1, ::-1]
    db = dyr.sum(0)
    return conv_adjoint(dy, W, (H, Wd)), np.ascontiguousarray(dW), db


def conv_adjoint(y, W, out_hw, crop=(0, 0)):
    """Adjoint of `conv_valid applied to an image zero-padded by crop`: y (B,O,OH,OW) -> (B,C,H,W) with
    H = OH + kh - 1 - 2 crop_h.  This is Deconv2DLayer(W=conv.W, flip_filters=False, crop=crop) without bias."""
    O, C, kh, kw = W.shape
    B, _, OH, OW = y.shape
    Hp, Wp = OH + kh - 1, OW + kw - 1
    Wf = W[:, :, ::-1, ::-1].reshape(O, -1)
    dcols = (y.transpose(0, 2, 3, 1).reshape(-1, O) @ Wf).reshape(B, OH, OW, C, kh, kw)
    out = np.zeros((B, C, Hp, Wp), dtype=y.dtype)
    for i in range(kh):
        for j in range(kw):
            out[:, :, i:i + OH, j:j + OW] += dcols[:, :, :, :, i, j].transpose(0, 3, 1, 2)
    ch, cw = crop
    out = out[:, :, ch:Hp - ch, cw:Wp - cw]
    assert out.shape[2:] == tuple(out_hw), (out.shape, out_hw)
    return out


def maxpool2(x, pad=(0, 0)):
    """2x2 / stride 2, ignore_border=True, padded cells never selected.  Returns (y, argmax code 0..3 as dy*2+dx)."""
    B, C, H, W = x.shape
    ph, pw = pad
    OH, OW = (H + 2 * ph - 2) // 2 + 1, (W + 2 * pw - 2) // 2 + 1
    xp = np.full((B, C, 2 * OH, 2 * OW), -np.inf, dtype=x.dtype)
    h_in, w_in = min(H, 2 * OH - ph), min(W, 2 * OW - pw)
    xp[:, :, ph:ph + h_in, pw:pw + w_in] = x[:, :, :h_in, :w_in]
    win = xp.reshape(B, C, OH, 2, OW, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, OH, OW, 4)
    arg = win.argmax(-1)
    return np.take_along_axis(win, arg[..., None], -1)[..., 0], arg.astype(np.uint8)


def maxpool2_bwd(dy, arg, in_hw, pad=(0, 0)):
    B, C, OH, OW = dy.shape
    ph, pw = pad
    H, W = in_hw
    g = np.zeros((B, C, OH, OW, 4), dtype=dy.dtype)
    np.put_along_axis(g, arg[..., None].astype(np.int64), dy[..., None], -1)
    gp = g.reshape(B, C, OH, OW, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, 2 * OH, 2 * OW)
    out = np.zeros((B, C, H, W), dtype=dy.dtype)
    h_in, w_in = min(H, 2 * OH - ph), min(W, 2 * OW - pw)
    out[:, :, :h_in, :w_in] = gp[:, :, ph:ph + h_in, pw:pw + w_in]
    return out


def upscale2(x):
    return x.repeat(2, axis=2).repeat(2, axis=3)


def upscale2_bwd(dy):
    B, C, H, W = dy.shape
    return dy.reshape(B, C, H // 2, 2, W // 2, 2).sum(axis=(3, 5))


# --------------------------------------------------------------------------- the auto-encoder
def param_names():
    """lasagne.layers.get_all_params(network, trainable=True) order: layers in topological order, W before b; tied
    weights belong to the layer that created them."""
    return ["conv2d1.W", "conv2d1.b", "conv2d3.W", "conv2d3.b", "conv2d5.W", "conv2d5.b", "dense7.W", "dense7.b",
            "bottleneck.W", "bottleneck.b", "dense8.b", "dense9.b", "deconv2d11.b", "deconv2d13.b", "deconv2d14.b"]


def geometry(image_hw=(30, 40)):
    h, w = image_hw
    g = dict(in_hw=(h, w))
    g["c1"] = (h - 4, w - 4)
    g["p2"] = ((g["c1"][0] - 2) // 2 + 1, (g["c1"][1] - 2) // 2 + 1)
    g["c3"] = (g["p2"][0] - 4, g["p2"][1] - 4)
    g["p4"] = ((g["c3"][0] + 2 - 2) // 2 + 1, (g["c3"][1] - 2) // 2 + 1)       # pad (1, 0)
    g["c5"] = (g["p4"][0] - 2, g["p4"][1] - 2)
    g["flat"] = FILTERS[2] * g["c5"][0] * g["c5"][1]
    g["d11"] = (g["c5"][0] + 2, g["c5"][1] + 2)
    g["u12"] = (2 * g["d11"][0], 2 * g["d11"][1])
    g["d13"] = (g["u12"][0] + 4, g["u12"][1] + 4)
    g["u14"] = (2 * g["d13"][0], 2 * g["d13"][1])
    g["d15"] = (g["u14"][0] + 4 - 2, g["u14"][1] + 4)                           # crop (1, 0)
    return g


def param_shapes(dense=500, bottleneck=50, image_hw=(30, 40)):
    g = geometry(image_hw)
    return {"conv2d1.W": (FILTERS[0], 1, 5, 5), "conv2d1.b": (FILTERS[0],),
            "conv2d3.W": (FILTERS[1], FILTERS[0], 5, 5), "conv2d3.b": (FILTERS[1],),
            "conv2d5.W": (FILTERS[2], FILTERS[1], 3, 3), "conv2d5.b": (FILTERS[2],),
            "dense7.W": (g["flat"], dense), "dense7.b": (dense,),
            "bottleneck.W": (dense, bottleneck), "bottleneck.b": (bottleneck,),
            "dense8.b": (dense,), "dense9.b": (g["flat"],),
            "deconv2d11.b": (FILTERS[1],), "deconv2d13.b": (FILTERS[0],), "deconv2d14.b": (1,)}


def init_params(rng, dtype=np.float32, dense=500, bottleneck=50, image_hw=(30, 40), bias_noise=0.0):
    """GlorotUniform weights (Lasagne's default for conv and dense layers), zero biases (+ optional noise for tests)."""
    p = {}
    for k, shp in param_shapes(dense, bottleneck, image_hw).items():
        if k.endswith(".W"):
            if len(shp) == 4:
                fan_in, fan_out = shp[1] * shp[2] * shp[3], shp[0] * shp[2] * shp[3]
            else:
                fan_in, fan_out = shp
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            p[k] = rng.uniform(-lim, lim, shp).astype(dtype)
        else:
            p[k] = (rng.normal(0, bias_noise, shp) if bias_noise else np.zeros(shp)).astype(dtype)
    return p


def forward(p, x, image_hw=(30, 40), want_cache=False):
    """x (B, H*W) -> (reconstruction (B, H*W), code (B, bottleneck)) [, cache]."""
    B = x.shape[0]
    g = geometry(image_hw)
    c = dict(x0=x.reshape(B, 1, *image_hw))
    c["a1"] = stanh(conv_valid(c["x0"], p["conv2d1.W"], p["conv2d1.b"]))
    c["p2"], c["arg2"] = maxpool2(c["a1"])
    c["a3"] = stanh(conv_valid(c["p2"], p["conv2d3.W"], p["conv2d3.b"]))
    c["p4"], c["arg4"] = maxpool2(c["a3"], pad=(1, 0))
    c["a5"] = stanh(conv_valid(c["p4"], p["conv2d5.W"], p["conv2d5.b"]))
    c["f6"] = c["a5"].reshape(B, -1)
    c["a7"] = stanh(c["f6"] @ p["dense7.W"] + p["dense7.b"])
    c["code"] = c["a7"] @ p["bottleneck.W"] + p["bottleneck.b"]
    c["a8"] = c["code"] @ p["bottleneck.W"].T + p["dense8.b"]
    c["a9"] = stanh(c["a8"] @ p["dense7.W"].T + p["dense9.b"])
    c["r10"] = c["a9"].reshape(B, FILTERS[2], *g["c5"])
    c["a11"] = stanh(conv_adjoint(c["r10"], p["conv2d5.W"], g["d11"]) + p["deconv2d11.b"][None, :, None, None])
    c["u12"] = upscale2(c["a11"])
    c["a13"] = stanh(conv_adjoint(c["u12"], p["conv2d3.W"], g["d13"]) + p["deconv2d13.b"][None, :, None, None])
    c["u14"] = upscale2(c["a13"])
    c["a15"] = stanh(conv_adjoint(c["u14"], p["conv2d1.W"], g["d15"], crop=(1, 0)) + p["deconv2d14.b"][None, :, None, None])
    recon = c["a15"].reshape(B, -1)
    c["recon"] = recon
    return (recon, c["code"], c) if want_cache else (recon, c["code"])


def _adjoint_bwd(y_in, W, d_out, crop=(0, 0)):
    """Backward of z = conv_adjoint(y_in, W, ., crop): returns (d y_in, dW).  The adjoint of the adjoint is the
    convolution of the (crop-padded) gradient; the weight gradient is the convolution's with the roles swapped."""
    ch, cw = crop
    dp = np.pad(d_out, ((0, 0), (0, 0), (ch, ch), (cw, cw)))
    O, C, kh, kw = W.shape
    dy_in = conv_valid(dp, W, np.zeros(O, dtype=W.dtype))
    _, dW, _ = conv_valid_bwd(dp, W, y_in)
    return dy_in, dW


def loss_and_grads(p, x, target=None, image_hw=(30, 40)):
    """cost = mean((recon - target)^2) over all B*H*W elements (avletters/avletters_convae.py:256)."""
    target = x if target is None else target
    g = geometry(image_hw)
    recon, code, c = forward(p, x, image_hw, want_cache=True)
    n = recon.size
    loss = ((recon - target) ** 2).sum() / n
    gr = {k: np.zeros_like(v) for k, v in p.items()}
    B = x.shape[0]
    d = (2.0 / n) * (recon - target)
    d15 = d.reshape(B, 1, *g["d15"]) * stanh_grad_from_output(c["a15"])
    gr["deconv2d14.b"] += d15.sum((0, 2, 3))
    du14, dW = _adjoint_bwd(c["u14"], p["conv2d1.W"], d15, crop=(1, 0))
    gr["conv2d1.W"] += dW
    d13 = upscale2_bwd(du14) * stanh_grad_from_output(c["a13"])
    gr["deconv2d13.b"] += d13.sum((0, 2, 3))
    du12, dW = _adjoint_bwd(c["u12"], p["conv2d3.W"], d13)
    gr["conv2d3.W"] += dW
    d11 = upscale2_bwd(du12) * stanh_grad_from_output(c["a11"])
    gr["deconv2d11.b"] += d11.sum((0, 2, 3))
    dr10, dW = _adjoint_bwd(c["r10"], p["conv2d5.W"], d11)
    gr["conv2d5.W"] += dW
    d9 = dr10.reshape(B, -1) * stanh_grad_from_output(c["a9"])
    gr["dense9.b"] += d9.sum(0)
    gr["dense7.W"] += d9.T @ c["a8"]                        # a9_pre = a8 @ W7^T
    d8 = d9 @ p["dense7.W"]
    gr["dense8.b"] += d8.sum(0)
    gr["bottleneck.W"] += d8.T @ c["code"]                  # a8 = code @ Wb^T
    dcode = d8 @ p["bottleneck.W"]
    gr["bottleneck.b"] += dcode.sum(0)
    gr["bottleneck.W"] += c["a7"].T @ dcode
    d7 = (dcode @ p["bottleneck.W"].T) * stanh_grad_from_output(c["a7"])
    gr["dense7.b"] += d7.sum(0)
    gr["dense7.W"] += c["f6"].T @ d7
    d5 = (d7 @ p["dense7.W"].T).reshape(c["a5"].shape) * stanh_grad_from_output(c["a5"])
    dp4, dW, db = conv_valid_bwd(c["p4"], p["conv2d5.W"], d5)
    gr["conv2d5.W"] += dW; gr["conv2d5.b"] += db
    d3 = maxpool2_bwd(dp4, c["arg4"], g["c3"], pad=(1, 0)) * stanh_grad_from_output(c["a3"])
    dp2, dW, db = conv_valid_bwd(c["p2"], p["conv2d3.W"], d3)
    gr["conv2d3.W"] += dW; gr["conv2d3.b"] += db
    d1 = maxpool2_bwd(dp2, c["arg2"], g["c1"]) * stanh_grad_from_output(c["a1"])
    _, dW, db = conv_valid_bwd(c["x0"], p["conv2d1.W"], d1)
    gr["conv2d1.W"] += dW; gr["conv2d1.b"] += db
    return loss, gr, c
