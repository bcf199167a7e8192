"""CPU oracle for the convolutional auto-encoder.  TEST INFRASTRUCTURE ONLY: only tests/,
__graft_entry__.smoke() and bench legs that time a CPU baseline may import it.

NumPy restatement of reference modelzoo/avletters_convae.py:33-69 (the 'normal' model), of its three variants
(modelzoo/avletters_convae_bn.py:33-74 'batchnorm', avletters_convae_drop.py:33-75 'dropout',
avletters_convae_bndrop.py:33-77 'bn+dropout'; the keys are avletters/avletters_convae.py:245-252's --model values) and of
the training step (avletters/avletters_convae.py:254-262: mean squared error of the reconstruction, lasagne.updates.adadelta):

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
  * BatchNormLayer (eps 1e-4, alpha 0.1): statistics over all axes but 1 (per channel of a 4-d tensor, per feature of
    a 2-d one); parameters beta, gamma (trainable), mean, inv_std (running averages, updated by every non-deterministic
    pass); DropoutLayer: rescale=True, p = 0.2 on the input, 0.5 elsewhere; masks = adenet_oracle.dropout_uniform over
    the element index of the layer's (B, C, H, W) / (B, F) tensor with layer ids 0..4.
Arrays are NCHW here, like the reference.
"""
from __future__ import annotations

import numpy as np

SCALE_IN, SCALE_OUT = 0.5, 2.4            # modelzoo/avletters_convae.py:7-26
FILTERS = (100, 150, 200)
KSIZE = (5, 5, 3)


BN_EPS, BN_ALPHA = 1e-4, 0.1                 # lasagne.layers.BatchNormLayer defaults
_NORMAL_NAMES = dict(c1="conv2d1", c3="conv2d3", c5="conv2d5", d7="dense7", d8="dense8", d9="dense9", dc11="deconv2d11",
                     dc13="deconv2d13", dc15="deconv2d14")
# bn: "pool" = BatchNormLayers behind the two poolings, on the flattened conv output (per FEATURE) and behind the dense
# layer (avletters_convae_bn.py:49-59); "conv" = behind every convolution's nonlinearity (per channel, also the third)
# and behind the dense layer (avletters_convae_bndrop.py:48-62)
VARIANTS = {
    "normal": dict(filters=(100, 150, 200), tanh=(0.5, 2.4), bn=None, drop=False, widen=1, names=_NORMAL_NAMES),
    "batchnorm": dict(filters=(100, 150, 200), tanh=(0.5, 2.4), bn="pool", drop=False, widen=1,
                      bn_names=("batchnorm2", "batchnorm3", "batchnorm8", "batchnorm11"),
                      names=dict(c1="conv2d1", c3="conv2d4", c5="conv2d7", d7="dense10", d8="dense12", d9="dense13",
                                 dc11="deconv2d19", dc13="deconv2d17", dc15="deconv2d14")),
    # int(100 / 0.8), int(150 / 0.5), int(200 / 0.5); DENSE and BOTTLENECK doubled (avletters_convae_drop.py:34-44)
    "dropout": dict(filters=(125, 300, 400), tanh=(0.5, 2.4), bn=None, drop=True, widen=2, names=_NORMAL_NAMES),
    "bn+dropout": dict(filters=(100, 150, 200), tanh=(2. / 3, 1.7159), bn="conv", drop=True, widen=1,
                       bn_names=("batchnorm1", "batchnorm2", "batchnorm3", "batchnorm4"), names=_NORMAL_NAMES),
}
DROP_P = (0.2, 0.5, 0.5, 0.5, 0.5)           # dropout0 .. dropout4


def stanh(x, scale=(SCALE_IN, SCALE_OUT)):
    return x.dtype.type(scale[1]) * np.tanh(x.dtype.type(scale[0]) * x)


def stanh_grad_from_output(y, scale=(SCALE_IN, SCALE_OUT)):
    """d/dx [so tanh(si x)] = si so (1 - (y / so)^2); 1.2 (1 - (y / 2.4)^2) for ScaledTanh(0.5, 2.4)."""
    t = y / y.dtype.type(scale[1])
    return y.dtype.type(scale[0] * scale[1]) * (1 - t * t)


def bn_forward(x, p, name, training):
    """x (B, C, H, W) or (B, F): statistics over every axis but 1.  Returns (y, cache)."""
    axes = (0,) + tuple(range(2, x.ndim))
    shp = (1, -1) + (1,) * (x.ndim - 2)
    if training:
        mean = x.mean(axes)
        inv = 1.0 / np.sqrt(((x - mean.reshape(shp)) ** 2).mean(axes) + x.dtype.type(BN_EPS))
    else:
        mean, inv = p[name + ".mean"], p[name + ".inv_std"]
    xhat = (x - mean.reshape(shp)) * inv.reshape(shp)
    y = xhat * p[name + ".gamma"].reshape(shp) + p[name + ".beta"].reshape(shp)
    return y, dict(xhat=xhat, mean=mean, inv_std=inv, training=training, axes=axes, shp=shp)


def bn_backward(dy, p, name, c, gr):
    axes, shp = c["axes"], c["shp"]
    gr[name + ".beta"] += dy.sum(axes)
    gr[name + ".gamma"] += (dy * c["xhat"]).sum(axes)
    g = p[name + ".gamma"].reshape(shp) * c["inv_std"].reshape(shp)
    if not c["training"]:
        return dy * g
    n = dy.size // dy.shape[1]
    return g * (dy - dy.sum(axes).reshape(shp) / n - c["xhat"] * (dy * c["xhat"]).sum(axes).reshape(shp) / n)


def drop_scale(shape, layer, dropout, dtype):
    """mask / (1 - p) of DropoutLayer `layer` (0..4) for a tensor of `shape`; ones when deterministic."""
    from . import adenet_oracle as A
    if dropout is None:
        return np.ones(shape, dtype)
    p = DROP_P[layer]
    idx = np.arange(int(np.prod(shape)), dtype=np.uint64).reshape(shape)
    keep = A.dropout_uniform(dropout["seed"], dropout.get("counter", 0), layer, idx) >= np.float32(p)
    return keep.astype(dtype) * (dtype(1) / (dtype(1) - dtype(p)))



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
def param_names(variant="normal"):
    """lasagne.layers.get_all_params(network) order: layers in topological order, W before b, BatchNorm beta, gamma,
    mean, inv_std; tied weights belong to the layer that created them."""
    v = VARIANTS[variant]
    n = v["names"]
    bn = lambda k: [v["bn_names"][k] + s for s in (".beta", ".gamma", ".mean", ".inv_std")] if v["bn"] else []
    out = [n["c1"] + ".W", n["c1"] + ".b"] + bn(0) + [n["c3"] + ".W", n["c3"] + ".b"] + bn(1) + \
          [n["c5"] + ".W", n["c5"] + ".b"] + bn(2) + [n["d7"] + ".W", n["d7"] + ".b"] + bn(3) + \
          ["bottleneck.W", "bottleneck.b", n["d8"] + ".b", n["d9"] + ".b", n["dc11"] + ".b", n["dc13"] + ".b", n["dc15"] + ".b"]
    return out


def trainable_names(variant="normal"):
    return [k for k in param_names(variant) if not k.endswith((".mean", ".inv_std"))]


def geometry(image_hw=(30, 40), variant="normal"):
    h, w = image_hw
    F3 = VARIANTS[variant]["filters"][2]
    g = dict(in_hw=(h, w))
    g["c1"] = (h - 4, w - 4)
    g["p2"] = ((g["c1"][0] - 2) // 2 + 1, (g["c1"][1] - 2) // 2 + 1)
    g["c3"] = (g["p2"][0] - 4, g["p2"][1] - 4)
    g["p4"] = ((g["c3"][0] + 2 - 2) // 2 + 1, (g["c3"][1] - 2) // 2 + 1)       # pad (1, 0)
    g["c5"] = (g["p4"][0] - 2, g["p4"][1] - 2)
    g["flat"] = F3 * g["c5"][0] * g["c5"][1]
    g["d11"] = (g["c5"][0] + 2, g["c5"][1] + 2)
    g["u12"] = (2 * g["d11"][0], 2 * g["d11"][1])
    g["d13"] = (g["u12"][0] + 4, g["u12"][1] + 4)
    g["u14"] = (2 * g["d13"][0], 2 * g["d13"][1])
    g["d15"] = (g["u14"][0] + 4 - 2, g["u14"][1] + 4)                           # crop (1, 0)
    return g


def param_shapes(dense=500, bottleneck=50, image_hw=(30, 40), variant="normal"):
    """dense / bottleneck: the layer widths as built (the 'dropout' factory doubles options['DENSE'/'BOTTLENECK'])."""
    v = VARIANTS[variant]
    g = geometry(image_hw, variant)
    n, (F1, F2, F3) = v["names"], v["filters"]
    shp = {n["c1"] + ".W": (F1, 1, 5, 5), n["c1"] + ".b": (F1,),
           n["c3"] + ".W": (F2, F1, 5, 5), n["c3"] + ".b": (F2,),
           n["c5"] + ".W": (F3, F2, 3, 3), n["c5"] + ".b": (F3,),
           n["d7"] + ".W": (g["flat"], dense), n["d7"] + ".b": (dense,),
           "bottleneck.W": (dense, bottleneck), "bottleneck.b": (bottleneck,),
           n["d8"] + ".b": (dense,), n["d9"] + ".b": (g["flat"],),
           n["dc11"] + ".b": (F2,), n["dc13"] + ".b": (F1,), n["dc15"] + ".b": (1,)}
    if v["bn"]:
        widths = (F1, F2, g["flat"] if v["bn"] == "pool" else F3, dense)
        for name, wd in zip(v["bn_names"], widths):
            for s in (".beta", ".gamma", ".mean", ".inv_std"):
                shp[name + s] = (wd,)
    return {k: shp[k] for k in param_names(variant)}


def init_params(rng, dtype=np.float32, dense=500, bottleneck=50, image_hw=(30, 40), bias_noise=0.0, variant="normal"):
    """GlorotUniform weights (Lasagne's default for conv and dense layers), zero biases (+ optional noise for tests);
    BatchNorm: beta 0, gamma 1, mean 0, inv_std 1 (+ noise)."""
    p = {}
    for k, shp in param_shapes(dense, bottleneck, image_hw, variant).items():
        if k.endswith(".W"):
            if len(shp) == 4:
                fan_in, fan_out = shp[1] * shp[2] * shp[3], shp[0] * shp[2] * shp[3]
            else:
                fan_in, fan_out = shp
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            p[k] = rng.uniform(-lim, lim, shp).astype(dtype)
        elif k.endswith((".gamma", ".inv_std")):
            p[k] = (1.0 + (rng.uniform(-0.3, 0.3, shp) if bias_noise else np.zeros(shp))).astype(dtype)
        else:
            p[k] = (rng.normal(0, bias_noise, shp) if bias_noise else np.zeros(shp)).astype(dtype)
    return p


def forward(p, x, image_hw=(30, 40), want_cache=False, variant="normal", dropout=None, training=None):
    """x (B, H*W) -> (reconstruction (B, H*W), code (B, bottleneck)) [, cache].
    dropout: None = deterministic, dict(seed=, counter=) = the DropoutLayers of the 'dropout' / 'bn+dropout' variants
    active; training: BatchNormLayers use batch statistics (default: dropout is not None)."""
    v = VARIANTS[variant]
    n, sc, (F1, F2, F3) = v["names"], v["tanh"], v["filters"]
    if training is None:
        training = dropout is not None
    drop = dropout if v["drop"] else None
    B = x.shape[0]
    g = geometry(image_hw, variant)
    c = dict(x0=x.reshape(B, 1, *image_hw), training=training)
    dt = x.dtype.type

    def bn(k, a):
        y, c["bn%d" % k] = bn_forward(a, p, v["bn_names"][k], training)
        return y

    def dr(k, a):
        c["drop%d" % k] = drop_scale(a.shape, k, drop, dt)
        return a * c["drop%d" % k]

    c["in1"] = dr(0, c["x0"])
    c["a1"] = stanh(conv_valid(c["in1"], p[n["c1"] + ".W"], p[n["c1"] + ".b"]), sc)
    t = bn(0, c["a1"]) if v["bn"] == "conv" else c["a1"]
    c["p2"], c["arg2"] = maxpool2(t)
    t = bn(0, c["p2"]) if v["bn"] == "pool" else c["p2"]
    c["in3"] = dr(1, t)
    c["a3"] = stanh(conv_valid(c["in3"], p[n["c3"] + ".W"], p[n["c3"] + ".b"]), sc)
    t = bn(1, c["a3"]) if v["bn"] == "conv" else c["a3"]
    c["p4"], c["arg4"] = maxpool2(t, pad=(1, 0))
    t = bn(1, c["p4"]) if v["bn"] == "pool" else c["p4"]
    c["in5"] = dr(2, t)
    c["a5"] = stanh(conv_valid(c["in5"], p[n["c5"] + ".W"], p[n["c5"] + ".b"]), sc)
    t = bn(2, c["a5"]) if v["bn"] == "conv" else c["a5"]
    t = t.reshape(B, -1)
    t = bn(2, t) if v["bn"] == "pool" else t
    c["f6"] = dr(3, t)
    c["a7"] = stanh(c["f6"] @ p[n["d7"] + ".W"] + p[n["d7"] + ".b"], sc)
    t = bn(3, c["a7"]) if v["bn"] else c["a7"]
    c["in_b"] = dr(4, t)
    c["code"] = c["in_b"] @ p["bottleneck.W"] + p["bottleneck.b"]
    c["a8"] = c["code"] @ p["bottleneck.W"].T + p[n["d8"] + ".b"]
    c["a9"] = stanh(c["a8"] @ p[n["d7"] + ".W"].T + p[n["d9"] + ".b"], sc)
    c["r10"] = c["a9"].reshape(B, F3, *g["c5"])
    c["a11"] = stanh(conv_adjoint(c["r10"], p[n["c5"] + ".W"], g["d11"]) + p[n["dc11"] + ".b"][None, :, None, None], sc)
    c["u12"] = upscale2(c["a11"])
    c["a13"] = stanh(conv_adjoint(c["u12"], p[n["c3"] + ".W"], g["d13"]) + p[n["dc13"] + ".b"][None, :, None, None], sc)
    c["u14"] = upscale2(c["a13"])
    c["a15"] = stanh(conv_adjoint(c["u14"], p[n["c1"] + ".W"], g["d15"], crop=(1, 0)) + p[n["dc15"] + ".b"][None, :, None, None], sc)
    recon = c["a15"].reshape(B, -1)
    c["recon"] = recon
    return (recon, c["code"], c) if want_cache else (recon, c["code"])


def bn_running_update(p, cache, variant):
    """What Theano's default updates do with every non-deterministic pass (lasagne BatchNormLayer.get_output_for)."""
    v = VARIANTS[variant]
    if not v["bn"] or not cache["training"]:
        return
    for k, name in enumerate(v["bn_names"]):
        dt = p[name + ".mean"].dtype.type
        p[name + ".mean"] = (dt(1) - dt(BN_ALPHA)) * p[name + ".mean"] + dt(BN_ALPHA) * cache["bn%d" % k]["mean"].astype(dt)
        p[name + ".inv_std"] = (dt(1) - dt(BN_ALPHA)) * p[name + ".inv_std"] + dt(BN_ALPHA) * cache["bn%d" % k]["inv_std"].astype(dt)


def _adjoint_bwd(y_in, W, d_out, crop=(0, 0)):
    """Backward of z = conv_adjoint(y_in, W, ., crop): returns (d y_in, dW).  The adjoint of the adjoint is the
    convolution of the (crop-padded) gradient; the weight gradient is the convolution's with the roles swapped."""
    ch, cw = crop
    dp = np.pad(d_out, ((0, 0), (0, 0), (ch, ch), (cw, cw)))
    O, C, kh, kw = W.shape
    dy_in = conv_valid(dp, W, np.zeros(O, dtype=W.dtype))
    _, dW, _ = conv_valid_bwd(dp, W, y_in)
    return dy_in, dW


def loss_and_grads(p, x, target=None, image_hw=(30, 40), variant="normal", dropout=None, training=None):
    """cost = mean((recon - target)^2) over all B*H*W elements (avletters/avletters_convae.py:256).  Gradients of the
    BatchNorm running averages are zero (they are not trainable)."""
    target = x if target is None else target
    v = VARIANTS[variant]
    n, sc = v["names"], v["tanh"]
    g = geometry(image_hw, variant)
    recon, code, c = forward(p, x, image_hw, want_cache=True, variant=variant, dropout=dropout, training=training)
    sg = lambda y: stanh_grad_from_output(y, sc)
    cnt = recon.size
    loss = ((recon - target) ** 2).sum() / cnt
    gr = {k: np.zeros_like(v_) for k, v_ in p.items()}
    B = x.shape[0]

    def bnb(k, d):
        return bn_backward(d, p, v["bn_names"][k], c["bn%d" % k], gr)

    d = (2.0 / cnt) * (recon - target)
    d15 = d.reshape(B, 1, *g["d15"]) * sg(c["a15"])
    gr[n["dc15"] + ".b"] += d15.sum((0, 2, 3))
    du14, dW = _adjoint_bwd(c["u14"], p[n["c1"] + ".W"], d15, crop=(1, 0))
    gr[n["c1"] + ".W"] += dW
    d13 = upscale2_bwd(du14) * sg(c["a13"])
    gr[n["dc13"] + ".b"] += d13.sum((0, 2, 3))
    du12, dW = _adjoint_bwd(c["u12"], p[n["c3"] + ".W"], d13)
    gr[n["c3"] + ".W"] += dW
    d11 = upscale2_bwd(du12) * sg(c["a11"])
    gr[n["dc11"] + ".b"] += d11.sum((0, 2, 3))
    dr10, dW = _adjoint_bwd(c["r10"], p[n["c5"] + ".W"], d11)
    gr[n["c5"] + ".W"] += dW
    d9 = dr10.reshape(B, -1) * sg(c["a9"])
    gr[n["d9"] + ".b"] += d9.sum(0)
    gr[n["d7"] + ".W"] += d9.T @ c["a8"]                        # a9_pre = a8 @ W7^T
    d8 = d9 @ p[n["d7"] + ".W"]
    gr[n["d8"] + ".b"] += d8.sum(0)
    gr["bottleneck.W"] += d8.T @ c["code"]                  # a8 = code @ Wb^T
    dcode = d8 @ p["bottleneck.W"]
    gr["bottleneck.b"] += dcode.sum(0)
    gr["bottleneck.W"] += c["in_b"].T @ dcode
    t = (dcode @ p["bottleneck.W"].T) * c["drop4"]
    if v["bn"]:
        t = bnb(3, t)
    d7 = t * sg(c["a7"])
    gr[n["d7"] + ".b"] += d7.sum(0)
    gr[n["d7"] + ".W"] += c["f6"].T @ d7
    t = (d7 @ p[n["d7"] + ".W"].T) * c["drop3"]
    if v["bn"] == "pool":
        t = bnb(2, t)
    t = t.reshape(c["a5"].shape)
    if v["bn"] == "conv":
        t = bnb(2, t)
    d5 = t * sg(c["a5"])
    din5, dW, db = conv_valid_bwd(c["in5"], p[n["c5"] + ".W"], d5)
    gr[n["c5"] + ".W"] += dW; gr[n["c5"] + ".b"] += db
    t = din5 * c["drop2"]
    if v["bn"] == "pool":
        t = bnb(1, t)
    t = maxpool2_bwd(t, c["arg4"], g["c3"], pad=(1, 0))
    if v["bn"] == "conv":
        t = bnb(1, t)
    d3 = t * sg(c["a3"])
    din3, dW, db = conv_valid_bwd(c["in3"], p[n["c3"] + ".W"], d3)
    gr[n["c3"] + ".W"] += dW; gr[n["c3"] + ".b"] += db
    t = din3 * c["drop1"]
    if v["bn"] == "pool":
        t = bnb(0, t)
    t = maxpool2_bwd(t, c["arg2"], g["c1"])
    if v["bn"] == "conv":
        t = bnb(0, t)
    d1 = t * sg(c["a1"])
    _, dW, db = conv_valid_bwd(c["in1"], p[n["c1"] + ".W"], d1)
    gr[n["c1"] + ".W"] += dW; gr[n["c1"] + ".b"] += db
    return loss, gr, c
