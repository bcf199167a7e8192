"""RBM / DBN pre-training on the GPU (optional): the reference's MATLAB package ``dbn/*.m`` -- the offline
producer of the ``w1..wN / b1..bN`` .mat files the dense encoders are initialised from -- with its own function names:

    dbnParamsInit(type, hiddenActivationFunctions, hiddenLayers)        dbn/dbnParamsInit.m
    trainRBM(dataMatrix, dbnParams, numHid, layerType)                   dbn/trainRBM.m
    trainDBN(dataMatrix, dbnParams)                                      dbn/trainDBN.m
    RBMup(data, weights, hidbiases, hL_type)                             dbn/RBMup.m            (activations only)
    unfoldDBNtoAE(dbnParams, dbn, outputSize)                            dbn/unfoldDBNtoAE.m
    save_ae_mat(path, weights, biases)                                   what dbn/exampleDBN_AE.m leaves for the encoders

One minibatch = ``adn_rbm_train_batch`` (csrc/rbm.hip: five MFMA GEMMs + the sampling and update kernels).  MATLAB's random
streams cannot be reproduced; the noise is the library's counter-based hash (``seed``), the initial weights and the
per-epoch permutations come from ``rng`` (a ``numpy.random.RandomState``; default: the global one, like every other
initialiser of this package)."""
import ctypes as C

import numpy as np

from . import _lib

_TYPES = {"sigm": "sigmoid", "linear": "linear", "relu": "rectify", "tanh": "tanh", "leakyrelu": "leaky_rectify"}


def _act_code(layer_type):
    try:
        return _lib.ACT[_TYPES[layer_type.lower()]]
    except KeyError:
        raise ValueError("unknown layer type %r (sigm, linear, ReLu, tanh, leakyReLu)" % (layer_type,))


def dbnParamsInit(type=1, hiddenActivationFunctions=("sigm",), hiddenLayers=(100,)):
    rbm = dict(epochs=10, batchsize=100, lrW=0.1, lrVb=0.1, lrHb=0.1, lrW_linear=0.001, lrVb_linear=0.001, lrHb_linear=0.001,
               weightPenaltyL2=0.0002, initMomentum=0.5, finalMomentum=0.9, momentumEpochThres=5, type=1)
    return dict(rbmParams=rbm, type=type, inputActivationFunction="sigm",
                hiddenActivationFunctions=list(hiddenActivationFunctions), hiddenLayers=list(hiddenLayers))


class RBM(object):
    """One restricted Boltzmann machine on the device."""

    def __init__(self, num_vis, num_hid, layerType, dbnParams):
        r = dbnParams["rbmParams"]
        v_type, h_type = layerType
        lin = any(t.lower() in ("linear", "relu") for t in (v_type, h_type))         # dbn/trainRBM.m:47-51
        cfg = _lib.RbmConfig()
        cfg.num_vis, cfg.num_hid = int(num_vis), int(num_hid)
        cfg.vis_type, cfg.hid_type = _act_code(v_type), _act_code(h_type)
        cfg.cd_type, cfg.batchsize = int(r["type"]), int(r["batchsize"])
        cfg.lr_w = r["lrW_linear"] if lin else r["lrW"]
        cfg.lr_vb = r["lrVb_linear"] if lin else r["lrVb"]
        cfg.lr_hb = r["lrHb_linear"] if lin else r["lrHb"]
        cfg.weight_penalty = r["weightPenaltyL2"]
        self._lib = _lib.load()
        self._handle = C.c_void_p()
        _lib.check(self._lib.adn_rbm_create(C.byref(cfg), C.byref(self._handle)))
        self.num_vis, self.num_hid, self.layerType = int(num_vis), int(num_hid), (v_type, h_type)
        self._shapes = [(self.num_vis, self.num_hid), (self.num_hid,), (self.num_vis,)] * 2

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.adn_rbm_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get(self, which):
        """0 W, 1 hidbiases, 2 visbiases, 3..5 their momentum terms."""
        out = np.empty(self._shapes[which], dtype=np.float32)
        _lib.check(self._lib.adn_rbm_read(self._handle, which, out.ctypes.data_as(C.c_void_p)))
        return out

    def set(self, which, value):
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float32).reshape(self._shapes[which]))
        _lib.check(self._lib.adn_rbm_write(self._handle, which, v.ctypes.data_as(C.c_void_p)))

    def _data(self, data):
        if hasattr(data, "is_cuda"):
            import torch
            d = data.to(torch.float32).contiguous()
            _lib.check(self._lib.adn_rbm_set_stream(self._handle, C.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
            return d, C.c_void_p(d.data_ptr()), _lib.FLAG_DEVICE_INPUTS
        d = np.ascontiguousarray(np.asarray(data, dtype=np.float32))
        return d, d.ctypes.data_as(C.c_void_p), 0

    def train_batch(self, data, momentum, seed, counter, want_err=True):
        d, ptr, flags = self._data(data)
        if d.ndim != 2 or d.shape[1] != self.num_vis:
            raise ValueError("expected (n, %d) data" % self.num_vis)
        err = C.c_float()
        _lib.check(self._lib.adn_rbm_train_batch(self._handle, ptr, int(d.shape[0]), flags, float(momentum), int(seed) & 0xFFFFFFFF,
                                                 int(counter) & 0xFFFFFFFF, C.byref(err) if want_err else None))
        return float(err.value) if want_err else None

    def up(self, data):
        d, ptr, flags = self._data(data)
        out = np.empty((d.shape[0], self.num_hid), dtype=np.float32)
        _lib.check(self._lib.adn_rbm_up(self._handle, ptr, int(d.shape[0]), flags, out.ctypes.data_as(C.c_void_p)))
        return out


def trainRBM(dataMatrix, dbnParams, numHid, layerType, rng=None, seed=1234, verbose=True):
    """-> (rbm, errorPerBatch, errorPerSample); rbm = dict(W=, hidbiases=, visbiases=).  The error vectors hold, as in the
    reference (dbn/trainRBM.m:170-175), the LAST minibatch's squared error of every epoch over the number of batches / examples."""
    rng = rng or np.random
    r = dbnParams["rbmParams"]
    data = np.ascontiguousarray(np.asarray(dataMatrix, dtype=np.float32))
    n, dims = data.shape
    nb = -(-n // r["batchsize"])
    m = RBM(dims, numHid, layerType, dbnParams)
    std = 0.01 if "relu" in (layerType[0].lower(), layerType[1].lower()) else 0.1          # dbn/trainRBM.m:56-60
    m.set(0, std * rng.standard_normal((dims, numHid)))
    per_batch, per_sample, counter = [], [], 0
    for epoch in range(1, r["epochs"] + 1):
        if verbose:
            print("epoch = %d" % epoch)
        order = rng.permutation(n)
        momentum = r["finalMomentum"] if epoch > r["momentumEpochThres"] else r["initMomentum"]
        err = 0.0
        for b in range(nb):
            idx = order[b * r["batchsize"]:] if b == nb - 1 else order[b * r["batchsize"]:(b + 1) * r["batchsize"]]
            err = m.train_batch(data[idx], momentum, seed, counter, want_err=(b == nb - 1))
            counter += 1
        per_batch.append(err / nb)
        per_sample.append(err / n)
        if verbose:
            print("Mean Squared Error per sample = %g" % per_sample[-1])
            print("Mean Squared Error per Batch = %g" % per_batch[-1])
    rbm = dict(W=m.get(0), hidbiases=m.get(1), visbiases=m.get(2))
    m.close()
    return rbm, per_batch, per_sample


def RBMup(data, weights, hidbiases, hL_type):
    """Hidden activations (host arithmetic: a helper of trainDBN and of callers that inspect a trained stack)."""
    x = np.asarray(data, np.float32) @ np.asarray(weights, np.float32) + np.asarray(hidbiases, np.float32)
    t = hL_type.lower()
    if t == "sigm":
        return 1.0 / (1.0 + np.exp(-x))
    if t == "tanh":
        return np.tanh(x)
    if t == "relu":
        return np.maximum(0, x)
    if t == "leakyrelu":
        return np.maximum(0.01 * x, x)
    return x


def trainDBN(dataMatrix, dbnParams, rng=None, seed=1234, verbose=True):
    """Greedy layer-wise stacking (dbn/trainDBN.m:21-50) -> (dbn, errorPerBatch, errorPerSample); dbn = dict(W=[...],
    hidbiases=[...], visbiases=[...])."""
    acts = [dbnParams["inputActivationFunction"]] + list(dbnParams["hiddenActivationFunctions"])
    dbn = dict(W=[], hidbiases=[], visbiases=[])
    data = np.asarray(dataMatrix, dtype=np.float32)
    errs_b, errs_s = [], []
    for i, hid in enumerate(dbnParams["hiddenLayers"]):
        if verbose:
            print("Pretraining Layer %d with RBM: %d-%d " % (i + 1, data.shape[1], hid))
        rbm, eb, es = trainRBM(data, dbnParams, hid, (acts[i], acts[i + 1]), rng=rng, seed=seed + i, verbose=verbose)
        for k in ("W", "hidbiases", "visbiases"):
            dbn[k].append(rbm[k])
        errs_b.append(eb); errs_s.append(es)
        data = RBMup(data, rbm["W"], rbm["hidbiases"], acts[i + 1]).astype(np.float32)     # the hidden PROBABILITIES
    if verbose:
        print("DBN training done")
    return dbn, errs_b, errs_s


def unfoldDBNtoAE(dbnParams, dbn, outputSize):
    """-> (weightsAE, biasesAE, newActivationFunctions, newLayers) (dbn/unfoldDBNtoAE.m:28-57)."""
    L = len(dbnParams["hiddenLayers"])
    if dbn["W"][0].shape[0] != outputSize:
        raise ValueError("Input size is different that output size. In an AE they should have the same size")
    weights = list(dbn["W"]) + [dbn["W"][i].T.copy() for i in range(L - 1, -1, -1)]
    biases = list(dbn["hidbiases"]) + [dbn["visbiases"][i] for i in range(L - 1, -1, -1)]
    hid = list(dbnParams["hiddenActivationFunctions"])
    acts = hid + hid[:-1][::-1] + [dbnParams["inputActivationFunction"]]
    layers = list(dbnParams["hiddenLayers"]) + list(dbnParams["hiddenLayers"][:-1])[::-1] + [outputSize]
    return weights, biases, acts, layers


def save_ae_mat(path, weights, biases):
    """w1..wN (in x out) and b1..bN (1 x out): the file ``runners.nstream.load_decoder`` / ``runners.modal.load_dbn`` read."""
    import scipy.io as sio
    d = {}
    for i, (w, b) in enumerate(zip(weights, biases)):
        d["w%d" % (i + 1)] = np.asarray(w, dtype=np.float32)
        d["b%d" % (i + 1)] = np.asarray(b, dtype=np.float32).reshape(1, -1)
    sio.savemat(path, d)
