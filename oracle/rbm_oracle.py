"""CPU oracle for the RBM / DBN pre-trainer (optional).  TEST INFRASTRUCTURE ONLY: only tests/,
__graft_entry__.smoke() and bench legs that time a CPU baseline may import it.

NumPy restatement of the reference's MATLAB pre-trainer, the offline producer of the ``w1..wN / b1..bN`` .mat files the
encoders are initialised from:
    dbn/trainRBM.m:28-183   contrastive divergence (CD-1), momentum 0.5 -> 0.9 after 5 epochs, L2 2e-4, learning rate 0.1
                            (0.001 as soon as a linear / ReLu layer is involved), batch 100
    dbn/RBMup.m, RBMdown.m, computeActivations.m, computeStates.m   the two conditionals and their sampling rules
    dbn/trainDBN.m:21-50    greedy layer-wise stacking (the next layer is trained on the hidden PROBABILITIES)
    dbn/unfoldDBNtoAE.m:28-57   encoder weights + transposed decoder weights, hidden biases then visible biases in reverse
    dbn/dbnParamsInit.m:19-52   the defaults

PARITY STATUS: parity unpinned against MATLAB (no MATLAB / Octave here, and its random streams could not be reproduced
anyway).  What the GPU implementation is held to is THIS restatement with the same counter-based random numbers
(adenet_oracle.dropout_uniform: uniform from a 24-bit hash; normal = Box-Muller of two of them), i.e. both draw identical
noise; the restatement itself is checked on properties (tests/test_rbm_oracle.py): CD-1 statistics against a direct
evaluation, the momentum / learning-rate schedule, the unfolded auto-encoder reproducing RBMup / RBMdown.
"""
from __future__ import annotations

import numpy as np

from . import adenet_oracle as A

LAYER_TYPES = ("sigm", "tanh", "linear", "ReLu", "leakyReLu")


def dbn_params_init(type_=1, hidden_activation_functions=("sigm",), hidden_layers=(100,)):
    """dbn/dbnParamsInit.m:19-52."""
    rbm = dict(epochs=10, batchsize=100, lrW=0.1, lrVb=0.1, lrHb=0.1, lrW_linear=0.001, lrVb_linear=0.001,
               lrHb_linear=0.001, weightPenaltyL2=0.0002, initMomentum=0.5, finalMomentum=0.9, momentumEpochThres=5, type=1)
    return dict(rbmParams=rbm, type=type_, inputActivationFunction="sigm",
                hiddenActivationFunctions=list(hidden_activation_functions), hiddenLayers=list(hidden_layers))


def compute_activations(layer_type, x):
    """dbn/computeActivations.m (the forms a layer of an RBM can take)."""
    t = layer_type.lower()
    if t == "sigm":
        return 1.0 / (1.0 + np.exp(-x))
    if t == "tanh":
        return 2.0 * (1.0 / (1.0 + np.exp(-2.0 * x))) - 1.0
    if t == "linear":
        return x
    if t == "relu":
        return np.maximum(0, x)
    if t == "leakyrelu":
        return np.maximum(0.01 * x, x)
    raise ValueError(layer_type)


def uniform(rng, stream, shape):
    """rng = dict(seed=, counter=): uniform in [0, 1) with 24 bits, element index in C order."""
    idx = np.arange(int(np.prod(shape)), dtype=np.uint64).reshape(shape)
    return A.dropout_uniform(rng["seed"], rng["counter"], stream, idx)


def normal(rng, stream, shape, dtype):
    """Box-Muller of the streams (stream, stream + 1): sqrt(-2 ln u1) cos(2 pi u2), u1 = (h1 + 0.5) 2^-24 in (0, 1)."""
    u1 = (uniform(rng, stream, shape) * 16777216.0 + 0.5) / 16777216.0
    u2 = uniform(rng, stream + 1, shape)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(dtype)


def compute_states(layer_type, probs, x, rng, stream):
    """dbn/computeStates.m: sigm -> Bernoulli(probs); linear -> probs + N(0, 1); ReLu -> max(0, x + sigmoid(x) N(0, 1))."""
    t = layer_type.lower()
    if t == "sigm":
        return (probs > uniform(rng, stream, probs.shape).astype(probs.dtype)).astype(probs.dtype)
    if t == "linear":
        return probs + normal(rng, stream, probs.shape, probs.dtype)
    if t == "relu":
        return np.maximum(0, x + (1.0 / (1.0 + np.exp(-x))) * normal(rng, stream, probs.shape, probs.dtype))
    raise ValueError("no sampling rule for layer type %r (dbn/computeStates.m)" % layer_type)


def rbm_up(data, W, hidbiases, h_type, rng=None):
    x = data @ W + hidbiases
    p = compute_activations(h_type, x)
    return (p, compute_states(h_type, p, x, rng, 0)) if rng is not None else (p, None)


def rbm_down(states, W, visbiases, v_type, rng=None):
    x = states @ W.T + visbiases
    p = compute_activations(v_type, x)
    return (p, compute_states(v_type, p, x, rng, 2)) if rng is not None else (p, None)


def learning_rates(params, v_type, h_type):
    r = params["rbmParams"]
    if any(t.lower() in ("linear", "relu") for t in (v_type, h_type)):
        return r["lrW_linear"], r["lrVb_linear"], r["lrHb_linear"]
    return r["lrW"], r["lrVb"], r["lrHb"]


def init_rbm(num_vis, num_hid, v_type, h_type, rng_np, dtype=np.float64):
    std = 0.01 if "relu" in (v_type.lower(), h_type.lower()) else 0.1          # dbn/trainRBM.m:56-60
    return dict(W=(std * rng_np.standard_normal((num_vis, num_hid))).astype(dtype), hidbiases=np.zeros(num_hid, dtype),
                visbiases=np.zeros(num_vis, dtype), dW=np.zeros((num_vis, num_hid), dtype), dvis=np.zeros(num_vis, dtype),
                dhid=np.zeros(num_hid, dtype))


def cd1_batch(rbm, data, params, layer_type, momentum, rng):
    """One minibatch of dbn/trainRBM.m:98-160; returns the squared reconstruction error of the batch.
    rng = dict(seed=, counter=) selects the noise of this batch."""
    v_type, h_type = layer_type
    r = params["rbmParams"]
    lrW, lrVb, lrHb = learning_rates(params, v_type, h_type)
    W, n = rbm["W"], data.shape[0]
    pos_p, pos_s = rbm_up(data, W, rbm["hidbiases"], h_type, rng)
    pos_h = pos_p if r["type"] == 1 else pos_s
    posprods, poshidact, posvisact = data.T @ pos_h, pos_h.sum(0), data.sum(0)
    neg_vp, neg_vs = rbm_down(pos_s, W, rbm["visbiases"], v_type, rng if r["type"] == 2 else None)
    neg_v = neg_vp if r["type"] == 1 else neg_vs
    neg_hp, _ = rbm_up(neg_v, W, rbm["hidbiases"], h_type)
    negprods, negvisact, neghidact = neg_v.T @ neg_hp, neg_v.sum(0), neg_hp.sum(0)
    err = float(((data - neg_v) ** 2).sum())
    dt = W.dtype.type
    bs = dt(r["batchsize"])                               # (the reference divides by the NOMINAL batch size, also for a short last batch)
    rbm["dW"] = dt(momentum) * rbm["dW"] + dt(lrW) * ((posprods - negprods) / bs - dt(r["weightPenaltyL2"]) * W)
    rbm["dvis"] = dt(momentum) * rbm["dvis"] + dt(lrVb) * (posvisact - negvisact) / bs
    rbm["dhid"] = dt(momentum) * rbm["dhid"] + dt(lrHb) * (poshidact - neghidact) / bs
    rbm["W"] = W + rbm["dW"]
    rbm["visbiases"] = rbm["visbiases"] + rbm["dvis"]
    rbm["hidbiases"] = rbm["hidbiases"] + rbm["dhid"]
    return err


def train_rbm(data, params, num_hid, layer_type, rng_np, seed=1234, dtype=np.float64):
    """dbn/trainRBM.m as a whole: returns (rbm, errorPerBatch, errorPerSample) -- the two error vectors hold, like the
    reference's (trainRBM.m:170-175), the LAST minibatch's error of every epoch divided by the number of batches / examples."""
    r = params["rbmParams"]
    n = data.shape[0]
    nb = -(-n // r["batchsize"])
    rbm = init_rbm(data.shape[1], num_hid, layer_type[0], layer_type[1], rng_np, dtype)
    per_batch, per_sample, counter = [], [], 0
    for epoch in range(1, r["epochs"] + 1):
        order = rng_np.permutation(n)
        momentum = r["finalMomentum"] if epoch > r["momentumEpochThres"] else r["initMomentum"]
        err = 0.0
        for b in range(nb):
            idx = order[b * r["batchsize"]:] if b == nb - 1 else order[b * r["batchsize"]:(b + 1) * r["batchsize"]]
            err = cd1_batch(rbm, data[idx].astype(dtype), params, layer_type, momentum, dict(seed=seed, counter=counter))
            counter += 1
        per_batch.append(err / nb)
        per_sample.append(err / n)
    return rbm, per_batch, per_sample


def unfold_dbn_to_ae(params, dbn, output_size):
    """dbn/unfoldDBNtoAE.m:28-57 -> (weights, biases, activation functions, layer sizes) of the 2L-layer auto-encoder."""
    L = len(params["hiddenLayers"])
    if dbn["W"][0].shape[0] != output_size:
        raise ValueError("Input size is different that output size. In an AE they should have the same size")
    weights = list(dbn["W"]) + [dbn["W"][i].T for i in range(L - 1, -1, -1)]
    biases = list(dbn["hidbiases"]) + [dbn["visbiases"][i] for i in range(L - 1, -1, -1)]
    acts = list(params["hiddenActivationFunctions"]) + list(params["hiddenActivationFunctions"][:-1])[::-1] + \
        [params["inputActivationFunction"]]
    layers = list(params["hiddenLayers"]) + list(params["hiddenLayers"][:-1])[::-1] + [output_size]
    return weights, biases, acts, layers
