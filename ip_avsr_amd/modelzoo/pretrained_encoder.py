"""Encoder descriptions (reference modelzoo/pretrained_encoder.py:4-16).  In this package an encoder is not
a layer object but the 4-tuple every factory consumes: (weights, biases, shapes, nonlinearities)."""
import numpy as np

from .. import init as _init


def create_pretrained_encoder(incoming_dim, weights, biases, shapes, nonlinearities, names=None):
    n = len(shapes)
    return [np.asarray(w, "float32") for w in weights[:n]], [np.asarray(b, "float32").reshape(-1) for b in biases[:n]], \
        [int(s) for s in shapes], list(nonlinearities)


def create_encoder(incoming_dim, shapes, nonlinearities, names=None):
    """Randomly initialised encoder (DenseLayer defaults: GlorotUniform weights, zero biases)."""
    d, weights, biases = int(incoming_dim), [], []
    for u in shapes:
        weights.append(_init.GlorotUniform()((d, int(u))))
        biases.append(np.zeros((int(u),), "float32"))
        d = int(u)
    return weights, biases, [int(s) for s in shapes], list(nonlinearities)
