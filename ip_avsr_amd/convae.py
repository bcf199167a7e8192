"""Convolutional auto-encoder on the GPU: Python face of ``adn_cae_*`` (csrc/convae.hip).

Mirrors what ``avletters/avletters_convae.py:254-283`` builds around ``modelzoo/avletters_convae.create_model``:
``train`` (reconstruction + adadelta update), ``train_cost_fn`` / ``eval_cost_fn`` (mean squared error; the 'normal'
model has no stochastic layers, so they coincide), ``recon_fn`` and the encoder's bottleneck output.  Parameters are
exchanged in Lasagne's layouts and ``get_all_params`` order, so ``utils.io.save_model_params`` files interchange.
"""
import ctypes as C

import numpy as np

from . import _lib

PARAM, GRAD, STATE0, STATE1 = 0, 1, 2, 3
# --model of avletters/avletters_convae.py:245-252 -> adn_cae_variant, filters per convolution, names of the three
# convolutions in that model-zoo file
VARIANTS = {"normal": (0, (100, 150, 200), ("conv2d1", "conv2d3", "conv2d5")),
            "batchnorm": (1, (100, 150, 200), ("conv2d1", "conv2d4", "conv2d7")),
            "dropout": (2, (125, 300, 400), ("conv2d1", "conv2d3", "conv2d5")),
            "bn+dropout": (3, (100, 150, 200), ("conv2d1", "conv2d3", "conv2d5"))}


class ConvAE(object):
    def __init__(self, image_shape=(30, 40), dense=500, bottleneck=50, precision="f32", variant="normal"):
        """``dense`` / ``bottleneck``: the layer widths as built (modelzoo.avletters_convae_drop doubles the options)."""
        self._lib = _lib.load()
        cfg = _lib.CaeConfig()
        cfg.image_h, cfg.image_w = int(image_shape[0]), int(image_shape[1])
        cfg.dense, cfg.bottleneck = int(dense), int(bottleneck)
        cfg.precision = _lib.PRECISION[precision]
        cfg.variant, (f1, f2, f3), (n1, n3, n5) = VARIANTS[variant]
        self.variant = variant
        self.stochastic = variant in ("dropout", "bn+dropout")
        self.has_batchnorm = variant in ("batchnorm", "bn+dropout")
        self._handle = C.c_void_p()
        _lib.check(self._lib.adn_cae_create(C.byref(cfg), C.byref(self._handle)))
        self.image_shape = (cfg.image_h, cfg.image_w)
        self.D = cfg.image_h * cfg.image_w
        self.bottleneck = cfg.bottleneck
        g = _geometry(self.image_shape, f3)
        shapes = {n1 + ".W": (f1, 1, 5, 5), n3 + ".W": (f2, f1, 5, 5), n5 + ".W": (f3, f2, 3, 3)}
        self.param_names, self.param_shapes = [], {}
        info = _lib.ParamInfo()
        for i in range(self._lib.adn_cae_num_params(self._handle)):
            _lib.check(self._lib.adn_cae_param_info(self._handle, i, C.byref(info)))
            name = info.name.decode()
            self.param_names.append(name)
            self.param_shapes[name] = shapes.get(name, tuple(int(info.dims[k]) for k in range(info.ndim)))
        self.flat = g

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.adn_cae_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def _tensor(self, buffer, name):
        out = np.empty(self.param_shapes[name], dtype=np.float32)
        _lib.check(self._lib.adn_cae_read_tensor(self._handle, buffer, self.param_names.index(name),
                                                 out.ctypes.data_as(C.c_void_p)))
        return out

    def get_param(self, name):
        return self._tensor(PARAM, name)

    def get_grad(self, name):
        return self._tensor(GRAD, name)

    def set_param(self, name, value):
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float32))
        if v.shape != tuple(self.param_shapes[name]):
            raise ValueError("%s: expected shape %s, got %s" % (name, self.param_shapes[name], v.shape))
        _lib.check(self._lib.adn_cae_write_tensor(self._handle, PARAM, self.param_names.index(name),
                                                  v.ctypes.data_as(C.c_void_p)))

    def get_all_param_values(self):
        return [self.get_param(n) for n in self.param_names]

    def set_all_param_values(self, values):
        if len(values) != len(self.param_names):
            raise ValueError("expected %d arrays" % len(self.param_names))
        for n, v in zip(self.param_names, values):
            self.set_param(n, v)

    def set_params_dict(self, d):
        for n in self.param_names:
            self.set_param(n, d[n])

    def get_grads_dict(self):
        return {n: self.get_grad(n) for n in self.param_names}

    def init_params(self, rng=None):
        """Lasagne's defaults: GlorotUniform filters / weights, zero biases."""
        rng = rng or np.random
        for n in self.param_names:
            shp = self.param_shapes[n]
            if n.endswith(".W"):
                fan_in, fan_out = (shp[1] * shp[2] * shp[3], shp[0] * shp[2] * shp[3]) if len(shp) == 4 else shp
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                self.set_param(n, rng.uniform(-lim, lim, shp))
            elif n.endswith((".gamma", ".inv_std")):         # BatchNormLayer: beta 0, gamma 1, mean 0, inv_std 1
                self.set_param(n, np.ones(shp))
            else:
                self.set_param(n, np.zeros(shp))

    # ------------------------------------------------------------------ data plumbing
    def _prep(self, x, target=None):
        dev = hasattr(x, "is_cuda")
        keep = []

        def one(a):
            if dev:
                import torch
                a = a.to(torch.float32).reshape(-1, self.D).contiguous()
                keep.append(a)
                return a.data_ptr(), a.shape[0]
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, self.D))
            keep.append(a)
            return a.ctypes.data, a.shape[0]

        xp, B = one(x)
        tp = None
        if target is not None:
            tp, Bt = one(target)
            if Bt != B:
                raise ValueError("input and target batch sizes differ")
        if dev:
            import torch
            _lib.check(self._lib.adn_cae_set_stream(self._handle, C.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
        return xp, tp, B, (_lib.FLAG_DEVICE_INPUTS if dev else 0), keep

    # ------------------------------------------------------------------ the compiled functions of the script
    def recon_fn(self, x):
        """Reconstruction (B, H*W) float32 (avletters/avletters_convae.py:268)."""
        xp, _, B, flags, keep = self._prep(x)
        out = np.empty((B, self.D), dtype=np.float32)
        _lib.check(self._lib.adn_cae_forward(self._handle, xp, B, flags, out.ctypes.data_as(C.c_void_p), None))
        return out

    def encode(self, x):
        """Bottleneck features (B, bottleneck): the alternative feature extractor of the AVSR front-end."""
        xp, _, B, flags, keep = self._prep(x)
        out = np.empty((B, self.bottleneck), dtype=np.float32)
        _lib.check(self._lib.adn_cae_forward(self._handle, xp, B, flags, None, out.ctypes.data_as(C.c_void_p)))
        return out

    def encode_device(self, x):
        """Bottleneck features of a torch CUDA tensor (B, H*W) as a torch CUDA tensor (B, bottleneck): no host round trip."""
        import torch
        xp, _, B, flags, keep = self._prep(x)
        out = torch.empty((B, self.bottleneck), device=x.device, dtype=torch.float32)
        _lib.check(self._lib.adn_cae_forward(self._handle, xp, B, flags | _lib.FLAG_DEVICE_OUTPUTS, None,
                                             C.c_void_p(out.data_ptr())))
        return out

    def cost(self, x, target=None, deterministic=True):
        """eval_cost_fn (deterministic) / train_cost_fn (``deterministic=False``: dropout masks drawn, BatchNorm on batch
        statistics incl. the running-average update, as get_output(network, deterministic=False) does): mean squared
        reconstruction error."""
        xp, tp, B, flags, keep = self._prep(x, target)
        if not deterministic:
            flags |= _lib.FLAG_STOCHASTIC
        out = C.c_float()
        _lib.check(self._lib.adn_cae_loss(self._handle, xp, tp, B, flags, C.byref(out)))
        return np.float32(out.value)

    def set_dropout_state(self, seed, counter=0):
        """Masks are a hash of (seed, counter, layer, element); the counter advances with every non-deterministic pass."""
        _lib.check(self._lib.adn_cae_set_dropout_state(self._handle, int(seed) & 0xFFFFFFFF, int(counter) & 0xFFFFFFFF))

    def compute_grads(self, x, target=None, want_loss=True, deterministic=False):
        xp, tp, B, flags, keep = self._prep(x, target)
        if deterministic:
            flags |= _lib.FLAG_DETERMINISTIC
        out = C.c_float()
        _lib.check(self._lib.adn_cae_compute_grads(self._handle, xp, tp, B, flags, C.byref(out) if want_loss else None))
        return np.float32(out.value) if want_loss else None

    def apply_adadelta(self, learning_rate=0.8, rho=0.95, epsilon=1e-6):
        _lib.check(self._lib.adn_cae_apply_adadelta(self._handle, float(learning_rate), float(rho), float(epsilon)))

    def apply_adam(self, learning_rate=1e-3):
        _lib.check(self._lib.adn_cae_apply_adam(self._handle, float(learning_rate)))

    def train(self, x, target=None, learning_rate=0.8, want_loss=True):
        """One adadelta step on the reconstruction error (avletters/avletters_convae.py:257,262)."""
        loss = self.compute_grads(x, target, want_loss)
        self.apply_adadelta(learning_rate)
        return loss

    def synchronize(self):
        _lib.check(self._lib.adn_cae_synchronize(self._handle))


def _geometry(hw, f3=200):
    h, w = hw
    c1 = (h - 4, w - 4)
    p2 = ((c1[0] - 2) // 2 + 1, (c1[1] - 2) // 2 + 1)
    c3 = (p2[0] - 4, p2[1] - 4)
    p4 = (c3[0] // 2 + 1, (c3[1] - 2) // 2 + 1)
    c5 = (p4[0] - 2, p4[1] - 2)
    return f3 * c5[0] * c5[1]
