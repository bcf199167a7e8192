"""Weight initialisers selectable from the INI key ``weight_init`` (runners/3stream.py:198-206):
glorot -> GlorotUniform, norm -> Normal(0.1), uniform -> Uniform() = U(-0.01, 0.01), ortho -> Orthogonal().
Semantics follow lasagne.init [upstream].  The reference never seeds its RNG;
here the stream is explicit: ``set_rng(np.random.RandomState(seed))`` for reproducible runs."""
import numpy as np

_rng = np.random


def set_rng(rng):
    global _rng
    _rng = rng


def get_rng():
    return _rng


class Initializer(object):
    def __call__(self, shape):
        return self.sample(tuple(shape)).astype(np.float32)


class Constant(Initializer):
    def __init__(self, val=0.0):
        self.val = val

    def sample(self, shape):
        return np.full(shape, self.val, dtype=np.float64)


class Normal(Initializer):
    def __init__(self, std=0.01, mean=0.0):
        self.std, self.mean = std, mean

    def sample(self, shape):
        return _rng.normal(self.mean, self.std, size=shape)


class Uniform(Initializer):
    def __init__(self, range=0.01):
        self.range = (-range, range) if np.isscalar(range) else tuple(range)

    def sample(self, shape):
        return _rng.uniform(self.range[0], self.range[1], size=shape)


class GlorotUniform(Initializer):
    """U(+-gain*sqrt(6/(fan_in+fan_out)))."""

    def __init__(self, gain=1.0):
        self.gain = gain

    def sample(self, shape):
        if len(shape) < 2:
            raise RuntimeError("This initializer only works with shapes of length >= 2")
        n1, n2 = shape[:2]
        lim = self.gain * np.sqrt(6.0 / (n1 + n2))
        return _rng.uniform(-lim, lim, size=shape)


class Orthogonal(Initializer):
    """Orthonormal rows/columns from the SVD of a Gaussian matrix (Saxe et al.)."""

    def __init__(self, gain=1.0):
        self.gain = gain

    def sample(self, shape):
        if len(shape) < 2:
            raise RuntimeError("Only shapes of length 2 or more are supported.")
        a = _rng.normal(0.0, 1.0, (shape[0], int(np.prod(shape[1:]))))
        u, _, v = np.linalg.svd(a, full_matrices=False)
        q = u if u.shape == a.shape else v
        return self.gain * q.reshape(shape)


def select(name):
    """INI ``weight_init`` string -> initialiser (unknown strings fall back to GlorotUniform, like the
    chain of ``if`` statements in runners/3stream.py:198-206)."""
    return {"glorot": GlorotUniform(), "norm": Normal(0.1), "uniform": Uniform(),
            "ortho": Orthogonal()}.get(name, GlorotUniform())


def resolve(init):
    if init is None:
        return GlorotUniform()
    if isinstance(init, str):
        return select(init)
    if isinstance(init, type):          # deltanet_majority_vote's default passes the CLASS GlorotUniform
        return init()
    return init
