"""The network object the model zoo returns: the host-side mirror of what the reference's scripts do
with a Lasagne output layer --

    predictions / cost / updates / theano.function x4   runners/3stream.py:304-320
    get_all_params / get_all_param_values / set_...     runners/3stream.py:305,393,425
    save_model_params / load_model_params               utils/io.py:40-48

backed by one ``adn_model`` of libadenet_hip.so.  Inputs follow the reference's conventions: NumPy
arrays passed by reference, silently down-cast (``allow_input_downcast=True``: float64 -> float32,
uint8 labels -> int32); outputs are fresh NumPy arrays; calls block.  ``torch`` CUDA tensors are
accepted as well and are then used in place (no host round trip).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import AdenetError

GATES = ("ingate", "forgetgate", "cell", "outgate")


def _act_name(a):
    if isinstance(a, str):
        name = a
    else:
        name = getattr(a, "__name__", None) or type(a).__name__
    name = {"rectify": "rectify", "relu": "rectify", "linear": "linear", "identity": "linear",
            "sigmoid": "sigmoid", "tanh": "tanh", "leaky_rectify": "leaky_rectify",
            "very_leaky_rectify": "very_leaky_rectify"}.get(name)
    if name is None:
        raise ValueError("unsupported encoder nonlinearity %r (supported: %s)" % (a, sorted(_lib.ACT)))
    return name


class Param(object):
    """Handle of one trainable tensor (stands in for a Theano shared variable)."""

    def __init__(self, model, index, name, shape):
        self._model, self.index, self.name, self.shape = model, index, name, tuple(shape)

    def get_value(self, borrow=False):
        return self._model._read(_lib.BUF_PARAM, self.index)

    def set_value(self, value):
        self._model._write(_lib.BUF_PARAM, self.index, value)

    def __repr__(self):
        return "<Param %s %s>" % (self.name, self.shape)


class PlaneInput(object):
    """A stream input as its two bfloat16 planes hi = bf16(x), lo = bf16(x - hi) (device tensors of x's shape): the operand
    form of the bf16x3 / mixed arithmetic (ADN_FLAG_PLANE_INPUTS) -- a resident split kept this way costs the same bytes as
    float32 and saves the model its split pass over every batch.  ``PlaneInput.split(x)`` makes one from a float32 tensor."""

    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo):
        if tuple(hi.shape) != tuple(lo.shape) or str(hi.dtype) != "torch.bfloat16" or str(lo.dtype) != "torch.bfloat16":
            raise ValueError("PlaneInput: two bfloat16 tensors of one shape")
        self.hi, self.lo = hi.contiguous(), lo.contiguous()

    @classmethod
    def split(cls, x):
        import torch
        x = x.to(torch.float32)
        hi = x.to(torch.bfloat16)
        return cls(hi, (x - hi.to(torch.float32)).to(torch.bfloat16))

    @property
    def shape(self):
        return self.hi.shape

    @property
    def ndim(self):
        return self.hi.ndim

    @property
    def device(self):
        return self.hi.device

    def __len__(self):
        return len(self.hi)

    def __getitem__(self, idx):
        return PlaneInput(self.hi[idx], self.lo[idx])

    def float(self):
        import torch
        return self.hi.to(torch.float32) + self.lo.to(torch.float32)


class AdeNetModel(object):
    """S-stream AdeNet / DeltaNet graph on one MI355X.

    ``spec`` is a plain dict::

        streams      list of {input_dim, enc_names, enc_shapes, enc_acts, delta, lstm_names, peepholes[, dropout]
                     [, batchnorm, aux_dim]}
                     (lstm_names of length 2 = summed forward/backward pair; dropout = p of a DropoutLayer ahead of
                     the stream's LSTM, modelzoo/adenet_v3.py:112; batchnorm = name of a BatchNormLayer on the encoder
                     output; aux_dim = width of an auxiliary input concatenated behind the delta features, its array
                     follows the stream inputs: modelzoo/adenet_v1.py:82-87)
        stream_lstm_size   units of the stream LSTMs when smaller than lstm_size (adenet_v1.py:89,95)
        fusion       'none' | 'sum' | 'adasum' | 'concat' ; fuse_name (layer name of the merge layer)
        agg_names    [] | [name] | [forward_name, backward_name] ; agg_peepholes
        lstm_size, classes, softmax_name
        head         'frames' (default: softmax on every frame, temporal_softmax_loss) | 'last' (SliceLayer(-1) +
                     softmax, categorical cross-entropy: modelzoo/adenet_v3.py:180-186, deltanet.py:48-56)
        agg_dropout  p of the DropoutLayer on the fused tensor (adenet_v3.py:154)
    """

    def __init__(self, spec, stream=None):
        self.spec = spec
        self._lib = _lib.load()
        cfg = _lib.Config()
        S = len(spec["streams"])
        if not 1 <= S <= _lib.ADN_MAX_STREAMS:
            raise ValueError("between 1 and %d streams are supported" % _lib.ADN_MAX_STREAMS)
        cfg.n_streams = S
        for k, s in enumerate(spec["streams"]):
            sc = cfg.streams[k]
            sc.input_dim = int(s["input_dim"])
            sc.n_enc = len(s["enc_shapes"])
            if sc.n_enc > _lib.ADN_MAX_ENC_LAYERS:
                raise ValueError("at most %d encoder layers per stream" % _lib.ADN_MAX_ENC_LAYERS)
            for l, (u, a) in enumerate(zip(s["enc_shapes"], s["enc_acts"])):
                sc.enc_units[l] = int(u)
                sc.enc_act[l] = _lib.ACT[_act_name(a)]
            sc.use_delta = int(bool(s["delta"]))
            sc.bidirectional = int(len(s["lstm_names"]) == 2)
            sc.peepholes = int(bool(s["peepholes"]))
            sc.dropout_p = float(s.get("dropout", 0.0) or 0.0)
            sc.batchnorm = int(bool(s.get("batchnorm")))
            sc.aux_dim = int(s.get("aux_dim", 0) or 0)
        if spec["fusion"] not in _lib.FUSION:
            # modelzoo/adenet_v2.py:75 raises for an unknown fusiontype (as a TypeError, through a
            # bug in the raise statement itself); here it is a plain ValueError
            raise ValueError("Unsupported Fusion Type used!")
        cfg.fusion = _lib.FUSION[spec["fusion"]]
        cfg.agg = len(spec["agg_names"])
        cfg.agg_peepholes = int(bool(spec.get("agg_peepholes", False)))
        cfg.lstm_size = int(spec["lstm_size"])
        cfg.classes = int(spec["classes"])
        cfg.precision = _lib.PRECISION[spec.get("precision", "f32")]
        cfg.head = _lib.HEAD[spec.get("head", "frames")]
        cfg.agg_dropout_p = float(spec.get("agg_dropout", 0.0) or 0.0)
        units = int(spec.get("stream_lstm_size") or 0)
        cfg.stream_lstm_units = units if 0 < units < cfg.lstm_size else 0
        self.head = spec.get("head", "frames")
        self._handle = C.c_void_p()
        _lib.check(self._lib.adn_create(C.byref(cfg), C.byref(self._handle)))
        self.S, self.H, self.C = S, cfg.lstm_size, cfg.classes
        self.input_dims = [int(s["input_dim"]) for s in spec["streams"]]
        self.aux_dims = [int(s["aux_dim"]) for s in spec["streams"] if s.get("aux_dim")]
        self._build_param_table()
        self._torch_stream = None
        self._front = {}                      # stream index -> (conv encoder, frame width): frozen feature extractors
        if stream is not None:
            self.set_stream(stream)
        if spec.get("relu_grad_at_zero"):     # (the same key the oracle reads: 0.5 = Theano's rectifier at exactly zero)
            self.set_relu_grad_at_zero(spec["relu_grad_at_zero"])

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.adn_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        """``stream``: a raw hipStream_t (int) or a torch.cuda.Stream."""
        raw = getattr(stream, "cuda_stream", stream)
        self._torch_stream = stream if hasattr(stream, "cuda_stream") else None
        _lib.check(self._lib.adn_set_stream(self._handle, C.c_void_p(int(raw))))

    def set_precision(self, precision):
        """'f32' (exact, parity-grade), 'bf16x3' (fp32-grade products as three bf16 MFMA passes), 'mixed' (bf16x3 forward pass
        and recurrences, ONE bf16 product per GEMM of back-propagation) or 'bf16' (GEMM operands rounded to bf16, fp32
        accumulate)."""
        _lib.check(self._lib.adn_set_precision(self._handle, _lib.PRECISION[precision]))
        self.spec["precision"] = precision

    def set_batch_lengths(self, lengths):
        """Frame compaction (include/adenet.h, csrc/compact.hip): announce the utterance lengths of the NEXT call's batch -- with the
        promise that its padding frames are zero, as utils/datagen.py makes them -- so that the encoders run over the valid
        frames + one zero row instead of all B x T.  Used up by that call; ``None`` withdraws an announcement."""
        if lengths is None:
            _lib.check(self._lib.adn_set_batch_lengths(self._handle, None, 0))
            return
        lens = np.ascontiguousarray(np.asarray(lengths).reshape(-1), dtype=np.int32)
        _lib.check(self._lib.adn_set_batch_lengths(self._handle, lens.ctypes.data_as(C.POINTER(C.c_int32)), int(lens.size)))

    def compact_rows(self):
        """Rows of the encoder matrices in the last call: sum(len) + 1 when it ran compacted (set_batch_lengths, or lengths read
        off a host mask), else 0."""
        return int(self._lib.adn_get_compact_rows(self._handle))

    def set_relu_grad_at_zero(self, value):
        """0 (default) or 0.5: the rectifier's derivative at a pre-activation of exactly zero (Theano's 0.5 (x + |x|) gives 0.5 --
        every zero-padded frame of a zero-bias encoder sits there; include/adenet.h).  A parity switch: the encoders leave the
        specialised kernels."""
        _lib.check(self._lib.adn_set_relu_grad_at_zero(self._handle, float(value)))

    def set_auto_compaction(self, on):
        """Host arrays: whether the lengths are read off a (prefix) mask so that the call can run compacted without an announcement
        (default on; the device checks the padding frames it was sent before relying on them -- include/adenet.h)."""
        _lib.check(self._lib.adn_set_auto_compaction(self._handle, 1 if on else 0))

    def bucket_rows(self):
        """Rows of the time-major tensors in the last call when it ran over length buckets, else 0 (include/adenet.h)."""
        return int(self._lib.adn_get_bucket_rows(self._handle))

    def set_length_buckets(self, on):
        """Whether a compacted train step may keep its recurrent side in length buckets (default on; include/adenet.h): same loss and
        gradients, ~24 % fewer time-major rows at AVLetters' lengths."""
        _lib.check(self._lib.adn_set_length_buckets(self._handle, 1 if on else 0))

    def synchronize(self):
        _lib.check(self._lib.adn_synchronize(self._handle))

    # ------------------------------------------------------------------ parameters
    def _lasagne_name(self, cname):
        spec = self.spec
        head, leaf = cname.split(".", 1)
        if head.startswith("stream"):
            s = spec["streams"][int(head[6:])]
            sub, leaf2 = leaf.split(".", 1)
            if sub.startswith("enc"):
                return "%s.%s" % (s["enc_names"][int(sub[3:])], leaf2)
            if sub == "bn":
                return "%s.%s" % (s["batchnorm"], leaf2)
            return "%s.%s" % (s["lstm_names"][int(sub[4:])], leaf2)
        if head == "fuse":
            return "%s.%s" % (spec["fuse_name"], leaf)
        if head.startswith("agg"):
            return "%s.%s" % (spec["agg_names"][int(head[3:])], leaf)
        return "%s.%s" % (spec["softmax_name"], leaf)

    def _build_param_table(self):
        n = self._lib.adn_num_params(self._handle)
        self.params = []
        self.param_index = {}
        info = _lib.ParamInfo()
        for i in range(n):
            _lib.check(self._lib.adn_param_info(self._handle, i, C.byref(info)))
            shape = tuple(int(info.dims[d]) for d in range(info.ndim))
            name = self._lasagne_name(info.name.decode())
            p = Param(self, i, name, shape)
            self.params.append(p)
            self.param_index[name] = i

    def _read(self, buf, index):
        p = self.params[index]
        out = np.empty(p.shape, dtype=np.float32)
        _lib.check(self._lib.adn_read_tensor(self._handle, buf, index, out.ctypes.data_as(C.c_void_p)))
        return out

    def _write(self, buf, index, value):
        p = self.params[index]
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float32))
        if v.shape != p.shape:
            if v.size == int(np.prod(p.shape)) and (v.ndim <= 2):
                v = np.ascontiguousarray(v.reshape(p.shape))      # e.g. (1,H) biases from a .mat file
            else:
                raise ValueError("mismatch: parameter %s has shape %s, got %s" % (p.name, p.shape, v.shape))
        _lib.check(self._lib.adn_write_tensor(self._handle, buf, index, v.ctypes.data_as(C.c_void_p)))

    def get_all_params(self, trainable=None, **tags):
        """lasagne.layers.get_all_params: every parameter on this path is trainable (encoders included)
        except a BatchNormLayer's running ``mean`` / ``inv_std``, which ``trainable=True`` filters out (they stay in the
        plain list: ``get_all_param_values`` checkpoints carry them).  ``scaling_param=True`` selects the adasum
        coefficients."""
        if tags.get("scaling_param"):
            return [p for p in self.params if ".adacoeff" in p.name]
        if trainable:
            bn = {s["batchnorm"] for s in self.spec["streams"] if s.get("batchnorm")}
            return [p for p in self.params
                    if not (p.name.rsplit(".", 1)[0] in bn and p.name.rsplit(".", 1)[1] in ("mean", "inv_std"))]
        return list(self.params)

    def running_statistic_names(self):
        """Names of the non-trainable running statistics (BatchNormLayer ``mean`` / ``inv_std``)."""
        trainable = {p.name for p in self.get_all_params(trainable=True)}
        return [p.name for p in self.params if p.name not in trainable]

    def get_all_param_values(self, **tags):
        return [p.get_value() for p in self.get_all_params(**tags)]

    def set_all_param_values(self, values, **tags):
        params = self.get_all_params(**tags)
        if len(values) != len(params):
            raise ValueError("mismatch: got %d values to set %d parameters" % (len(values), len(params)))
        for p, v in zip(params, values):
            p.set_value(v)

    def get_param(self, name):
        return self._read(_lib.BUF_PARAM, self.param_index[name])

    def set_param(self, name, value):
        self._write(_lib.BUF_PARAM, self.param_index[name], value)

    def get_params_dict(self):
        return {p.name: p.get_value() for p in self.params}

    def set_params_dict(self, d):
        for name, v in d.items():
            self.set_param(name, v)

    def get_grads_dict(self):
        return {p.name: self._read(_lib.BUF_GRAD, p.index) for p in self.params}

    def get_adam_state(self):
        return dict(t=self._lib.adn_adam_step_count(self._handle),
                    m=[self._read(_lib.BUF_ADAM_M, p.index) for p in self.params],
                    v=[self._read(_lib.BUF_ADAM_V, p.index) for p in self.params])

    def set_adam_state(self, state):
        _lib.check(self._lib.adn_set_adam_step_count(self._handle, int(state["t"])))
        for p, m, v in zip(self.params, state["m"], state["v"]):
            self._write(_lib.BUF_ADAM_M, p.index, m)
            self._write(_lib.BUF_ADAM_V, p.index, v)

    def snapshot_params(self):
        """The current parameters as ONE device-side copy of the flat buffer (72 MB for the 3-stream model: microseconds in
        HBM, against a per-tensor download for ``get_all_param_values``): what the epoch drivers keep as "best parameters so
        far" (runners/3stream.py:393) -- only a run that saves its best model ever needs them on the host."""
        import torch
        from .parallel import wrap_flat_buffer
        self.synchronize()
        snap = wrap_flat_buffer(self, _lib.BUF_PARAM, read_only=True).clone()     # (does not dirty the derived copies)
        torch.cuda.current_stream().synchronize()
        return snap

    def restore_params(self, snapshot):
        """Writes a ``snapshot_params`` copy back (the bf16 copies / planes are re-derived on the next use)."""
        import torch
        from .parallel import wrap_flat_buffer
        self.synchronize()
        flat = wrap_flat_buffer(self, _lib.BUF_PARAM)          # (adn_flat_buffer marks the parameters as written)
        if flat.numel() != snapshot.numel():
            raise ValueError("mismatch: the snapshot belongs to another model")
        flat.copy_(snapshot)
        torch.cuda.current_stream().synchronize()

    def snapshot_state(self):
        """Parameters, Adam moments and step count as device-side copies (what ``restore_state`` needs to put the training
        state back exactly): lets a measurement run extra steps without leaving a trace (bench.py's local-step clock)."""
        import torch
        from .parallel import wrap_flat_buffer
        self.synchronize()
        bufs = [wrap_flat_buffer(self, w, read_only=True).clone() for w in (_lib.BUF_PARAM, _lib.BUF_ADAM_M, _lib.BUF_ADAM_V)]
        torch.cuda.current_stream().synchronize()
        return dict(buffers=bufs, t=self.adam_step_count())

    def restore_state(self, snap):
        import torch
        from .parallel import wrap_flat_buffer
        self.synchronize()
        for w, src in zip((_lib.BUF_PARAM, _lib.BUF_ADAM_M, _lib.BUF_ADAM_V), snap["buffers"]):
            wrap_flat_buffer(self, w).copy_(src)             # (BUF_PARAM through adn_flat_buffer: marks the parameters as written)
        torch.cuda.current_stream().synchronize()
        self.set_adam_step_count(snap["t"])

    def count_params(self):
        return int(self._lib.adn_total_param_count(self._handle))

    def flat_buffer(self, which=_lib.BUF_GRAD, read_only=False):
        """(device pointer, bytes) of a flat fp32 buffer; BUF_GRAD is what data-parallel ranks all-reduce.  ``read_only``: the
        caller will not write through the pointer (adn_flat_buffer_const: the parameters are not marked as written)."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        fn = self._lib.adn_flat_buffer_const if read_only else self._lib.adn_flat_buffer
        _lib.check(fn(self._handle, which, C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def grad_buckets(self):
        """[(begin, end)] float ranges of the flat gradient buffer in the order they become final during
        back-propagation: [fusion | aggregation | classifier | cost tail] first, then every stream's [BatchNorm | LSTM]
        range, then one [W_l | b_l] range per encoder layer (include/adenet.h: adn_grad_buckets)."""
        b = (C.c_int64 * 256)(); e = (C.c_int64 * 256)(); n = C.c_int()
        _lib.check(self._lib.adn_grad_buckets(self._handle, 256, b, e, C.byref(n)))
        return [(int(b[k]), int(e[k])) for k in range(n.value)]

    def grad_bucket_groups(self):
        """For every bucket of ``grad_buckets`` the number of its release point: buckets that share it (the streams' ranges
        behind one grouped launch) are consecutive and may be reduced by one grouped collective."""
        g = (C.c_int * 256)(); n = C.c_int()
        _lib.check(self._lib.adn_grad_bucket_groups(self._handle, 256, g, C.byref(n)))
        return [int(g[k]) for k in range(n.value)]

    def set_bucket_events(self, raw_events):
        """Raw hipEvent_t handles (ints), one per bucket, recorded by compute_grads; [] clears."""
        arr = (C.c_void_p * max(1, len(raw_events)))(*[C.c_void_p(int(h)) for h in raw_events])
        _lib.check(self._lib.adn_set_bucket_events(self._handle, arr, len(raw_events)))

    # ------------------------------------------------------------------ calls
    @staticmethod
    def _is_device(x):
        return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)

    def set_front_end(self, stream, conv_encoder, frame_dim):
        """Stream ``stream`` receives (B, T, frame_dim) frames; ``conv_encoder.encode`` turns them into the
        (B, T, input_dim) codes the graph consumes (a frozen convolutional feature extractor, csrc/convae.hip)."""
        if int(conv_encoder.bottleneck) != self.input_dims[stream]:
            raise ValueError("the encoder emits %d features, stream %d takes %d" % (conv_encoder.bottleneck, stream,
                                                                                    self.input_dims[stream]))
        self._front[int(stream)] = (conv_encoder, int(frame_dim))

    def _apply_front_ends(self, inputs):
        if not self._front:
            return inputs
        out = list(inputs)
        for k, (conv, fd) in self._front.items():
            x = out[k]
            if tuple(x.shape[2:]) != (fd,):
                raise ValueError("stream %d: expected (B,T,%d) frames, got %s" % (k, fd, tuple(x.shape)))
            B, T = int(x.shape[0]), int(x.shape[1])
            if self._is_device(x):
                out[k] = conv.encode_device(x.reshape(B * T, fd)).reshape(B, T, -1)
            else:
                out[k] = conv.encode(np.asarray(x, np.float32).reshape(B * T, fd)).reshape(B, T, -1)
        return out

    def _prep(self, inputs, mask, targets=None):
        n_in = self.S + len(self.aux_dims)
        if len(inputs) != n_in:
            raise ValueError("expected %d input streams%s, got %d arrays"
                             % (self.S, " + %d auxiliary inputs" % len(self.aux_dims) if self.aux_dims else "", len(inputs)))
        planes = all(isinstance(x, PlaneInput) for x in inputs)
        if planes:                                   # hi / lo planes of every stream (bf16x3 / mixed arithmetic)
            return self._prep_planes(inputs, mask, targets)
        if any(isinstance(x, PlaneInput) for x in inputs):
            raise ValueError("either every stream arrives as a PlaneInput or none")
        inputs = self._apply_front_ends(inputs)
        dev = self._is_device(inputs[0])
        # bfloat16 torch tensors (all inputs) go to the library as they are (ADN_FLAG_BF16_INPUTS): in bf16 mode the first
        # encoder GEMM reads a device array in place
        in16 = all(hasattr(x, "dtype") and str(x.dtype) == "torch.bfloat16" for x in inputs)
        keep = []
        ptrs = (C.c_void_p * n_in)()
        shape = None
        widths = self.input_dims + self.aux_dims
        for k, x in enumerate(inputs):
            if self._is_device(x) != dev:
                raise ValueError("all streams must live on the same side (host or device)")
            if in16:
                x = x.contiguous()
                p = x.data_ptr()
            elif dev:
                import torch
                if x.dtype != torch.float32 or not x.is_contiguous():
                    x = x.to(torch.float32).contiguous()
                p = x.data_ptr()
            else:
                x = np.ascontiguousarray(x, dtype=np.float32)
                p = x.ctypes.data
            if x.ndim != 3 or x.shape[2] != widths[k]:
                raise ValueError("input %d: expected (B,T,%d), got %s" % (k, widths[k], tuple(x.shape)))
            if shape is None:
                shape = tuple(x.shape[:2])
            elif tuple(x.shape[:2]) != shape:
                raise ValueError("streams disagree on (B,T): %s vs %s" % (tuple(x.shape[:2]), shape))
            keep.append(x)
            ptrs[k] = p
        B, T = shape

        def small(a, np_dtype, torch_name):
            if dev:
                import torch
                if getattr(a, "dev", None) is not None:        # utils/datagen_gpu.Resident: the HBM copy is at hand
                    a = a.dev
                if not self._is_device(a):
                    a = torch.as_tensor(np.ascontiguousarray(a, dtype=np_dtype), device=keep[0].device)
                a = a.to(getattr(torch, torch_name)).contiguous()
                if tuple(a.shape) != (B, T):
                    raise ValueError("mask/targets must be (B,T)=%s, got %s" % ((B, T), tuple(a.shape)))
                keep.append(a)
                return a.data_ptr()
            a = np.ascontiguousarray(a, dtype=np_dtype)
            if a.shape != (B, T):
                raise ValueError("mask/targets must be (B,T)=%s, got %s" % ((B, T), a.shape))
            keep.append(a)
            return a.ctypes.data

        mp = small(mask, np.uint8, "uint8")
        tp = small(targets, np.int32, "int32") if targets is not None else None
        if dev and self._torch_stream is None:
            import torch
            raw = torch.cuda.current_stream().cuda_stream
            _lib.check(self._lib.adn_set_stream(self._handle, C.c_void_p(int(raw))))
        flags = (_lib.FLAG_DEVICE_INPUTS if dev else 0) | (_lib.FLAG_BF16_INPUTS if in16 else 0)
        return ptrs, mp, tp, B, T, flags, keep

    def _prep_planes(self, inputs, mask, targets=None):
        import torch
        if self.aux_dims or self._front or self.spec.get("precision") not in ("bf16x3", "mixed"):
            # (another arithmetic, auxiliary inputs, a conv front end: the float32 values -- exact: hi + lo)
            return self._prep([x.float() for x in inputs], mask, targets)
        n = self.S
        ptrs = (C.c_void_p * (2 * n))()
        keep, shape = [], None
        for k, x in enumerate(inputs):
            if x.ndim != 3 or x.shape[2] != self.input_dims[k]:
                raise ValueError("input %d: expected (B,T,%d), got %s" % (k, self.input_dims[k], tuple(x.shape)))
            if shape is None:
                shape = tuple(x.shape[:2])
            elif tuple(x.shape[:2]) != shape:
                raise ValueError("streams disagree on (B,T): %s vs %s" % (tuple(x.shape[:2]), shape))
            keep += [x.hi, x.lo]
            ptrs[k], ptrs[n + k] = x.hi.data_ptr(), x.lo.data_ptr()
        B, T = shape

        def small(a, torch_dtype):
            if getattr(a, "dev", None) is not None:
                a = a.dev
            if not self._is_device(a):
                a = torch.as_tensor(np.ascontiguousarray(a), device=keep[0].device)
            a = a.to(torch_dtype).contiguous()
            if tuple(a.shape) != (B, T):
                raise ValueError("mask/targets must be (B,T)=%s, got %s" % ((B, T), tuple(a.shape)))
            keep.append(a)
            return a.data_ptr()

        mp = small(mask, torch.uint8)
        tp = small(targets, torch.int32) if targets is not None else None
        if self._torch_stream is None:
            _lib.check(self._lib.adn_set_stream(self._handle, C.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
        return ptrs, mp, tp, B, T, _lib.FLAG_DEVICE_INPUTS | _lib.FLAG_PLANE_INPUTS, keep

    def predict(self, inputs, mask, window):
        """val_fn (deterministic): probabilities (B,T,C) float32 -- (B,C) for the last-timestep head."""
        ptrs, mp, _, B, T, flags, keep = self._prep(inputs, mask)
        out = np.empty((B, self.C) if self.head == "last" else (B, T, self.C), dtype=np.float32)
        _lib.check(self._lib.adn_forward(self._handle, ptrs, mp, B, T, int(window), flags,
                                         out.ctypes.data_as(C.c_void_p)))
        return out

    def loss(self, inputs, targets, mask, window, deterministic=True):
        """compute_test_cost (deterministic=True) / compute_train_cost (False: dropout layers active, the
        reference's get_output(network, deterministic=False))."""
        ptrs, mp, tp, B, T, flags, keep = self._prep(inputs, mask, targets)
        if not deterministic:
            flags |= _lib.FLAG_STOCHASTIC
        out = C.c_float()
        _lib.check(self._lib.adn_loss(self._handle, ptrs, tp, mp, B, T, int(window), flags, C.byref(out)))
        return np.float32(out.value)

    def loss_and_probs(self, inputs, targets, mask, window):
        """compute_test_cost and val_fn of one batch from ONE forward pass: (cost, probabilities).  The reference's epoch loop
        calls the two compiled functions back to back on the held-out split (runners/3stream.py:373,383); both are
        deterministic passes over the same graph, so the second forward pass only recomputes what the first left behind."""
        ptrs, mp, tp, B, T, flags, keep = self._prep(inputs, mask, targets)
        out = C.c_float()
        _lib.check(self._lib.adn_loss(self._handle, ptrs, tp, mp, B, T, int(window), flags, C.byref(out)))
        probs = np.empty((B, self.C) if self.head == "last" else (B, T, self.C), dtype=np.float32)
        _lib.check(self._lib.adn_read_probs(self._handle, B, T, 0, probs.ctypes.data_as(C.c_void_p)))
        return np.float32(out.value), probs

    def set_dropout_state(self, seed, counter=0):
        """Dropout masks are a hash of (seed, counter, layer, element); the counter advances after every stochastic
        pass.  Setting both reproduces a draw (the oracle uses the same function)."""
        _lib.check(self._lib.adn_set_dropout_state(self._handle, int(seed) & 0xFFFFFFFF, int(counter) & 0xFFFFFFFF))

    def compute_grads(self, inputs, targets, mask, window, total_frames=0.0, want_loss=True, deterministic=False):
        ptrs, mp, tp, B, T, flags, keep = self._prep(inputs, mask, targets)
        if deterministic:
            flags |= _lib.FLAG_DETERMINISTIC
        out = C.c_float()
        _lib.check(self._lib.adn_compute_grads(self._handle, ptrs, tp, mp, B, T, int(window), flags,
                                               float(total_frames), C.byref(out) if want_loss else None))
        return np.float32(out.value) if want_loss else None

    def zero_grads(self):
        """``compute_grads`` of an EMPTY shard (data parallel: this rank got no utterance of a short minibatch): zero
        gradients and cost share, bucket events recorded, so that the rank still joins every all-reduce."""
        _lib.check(self._lib.adn_zero_grads(self._handle))

    def adam_step_count(self):
        return int(self._lib.adn_adam_step_count(self._handle))

    def set_adam_step_count(self, t):
        _lib.check(self._lib.adn_set_adam_step_count(self._handle, int(t)))

    def apply_adam(self, learning_rate):
        _lib.check(self._lib.adn_apply_adam(self._handle, float(learning_rate)))

    def adam_begin(self, learning_rate):
        """Opens an Adam step that is applied range by range (``adam_range``) and closed with ``adam_end``; covering
        every parameter once equals ``apply_adam`` bit for bit (data parallel: per-bucket updates)."""
        _lib.check(self._lib.adn_adam_begin(self._handle, float(learning_rate)))

    def adam_range(self, begin, end):
        _lib.check(self._lib.adn_adam_range(self._handle, int(begin), int(end)))

    def adam_ranges(self, ranges):
        """``adam_range`` for a list of (begin, end) ranges with one launch per 16 of them."""
        n = len(ranges)
        b = (C.c_int64 * max(1, n))(*[int(r[0]) for r in ranges]); e = (C.c_int64 * max(1, n))(*[int(r[1]) for r in ranges])
        _lib.check(self._lib.adn_adam_ranges(self._handle, b, e, n))

    def adam_end(self):
        _lib.check(self._lib.adn_adam_end(self._handle))

    def apply_sgd(self, learning_rate, momentum=0.0, nesterov=False):
        """lasagne.updates.sgd / momentum / nesterov_momentum on the gradients of the last compute_grads."""
        _lib.check(self._lib.adn_apply_sgd(self._handle, float(learning_rate), float(momentum), int(bool(nesterov))))

    def apply_adadelta(self, learning_rate=1.0, rho=0.95, epsilon=1e-6):
        """lasagne.updates.adadelta on the gradients of the last compute_grads."""
        _lib.check(self._lib.adn_apply_adadelta(self._handle, float(learning_rate), float(rho), float(epsilon)))

    def apply_adam_vlr(self, lr_map, default=None):
        """Adam with per-layer learning rates: ``lr_map`` maps Param handles or parameter names to rates
        (what ``custom.updates.generate_lr_map`` returns)."""
        rates = (C.c_float * len(self.params))()
        by_name = {(k.name if isinstance(k, Param) else k): float(v) for k, v in lr_map.items()}
        for i, p in enumerate(self.params):
            if p.name in by_name:
                rates[i] = by_name[p.name]
            elif default is not None:
                rates[i] = float(default)
            else:
                raise KeyError("no learning rate for parameter %s" % p.name)
        _lib.check(self._lib.adn_apply_adam_vlr(self._handle, rates, len(self.params)))

    def train_step(self, inputs, targets, mask, window, learning_rate, want_loss=True):
        """train(...): forward + backward + Adam; returns the cost of this batch before the update."""
        ptrs, mp, tp, B, T, flags, keep = self._prep(inputs, mask, targets)
        out = C.c_float()
        _lib.check(self._lib.adn_train_step(self._handle, ptrs, tp, mp, B, T, int(window), flags,
                                            float(learning_rate), C.byref(out) if want_loss else None))
        return np.float32(out.value) if want_loss else None

    def profile(self, on=True):
        """Start (on=True, counters cleared) or stop per-kernel-class HIP-event timing."""
        _lib.check(self._lib.adn_profile_enable(self._handle, int(bool(on))))

    def profile_read(self):
        """{class: dict(launches, ms, flops, bytes)} accumulated since profile(True); synchronises."""
        buf = (_lib.ProfileEntry * 16)()
        n = C.c_int()
        _lib.check(self._lib.adn_profile_read(self._handle, buf, 16, C.byref(n)))
        return {buf[i].name.decode(): dict(launches=int(buf[i].launches), ms=float(buf[i].ms),
                                           flops=float(buf[i].flops), bytes=float(buf[i].bytes))
                for i in range(n.value)}

    def encoder_activation(self, stream, layer, B, T):
        u = self.spec["streams"][stream]["enc_shapes"][layer]
        out = np.empty((B * T, u), dtype=np.float32)
        _lib.check(self._lib.adn_read_encoder_activation(self._handle, stream, layer, out.ctypes.data_as(C.c_void_p)))
        return out

    # ------------------------------------------------------------------ the four "compiled functions"
    def compile(self, learning_rate, order="inputs,targets,mask,window", updates="adam", **update_args):
        """Returns (train, compute_train_cost, compute_test_cost, val_fn) taking positional arguments in
        the order the reference script compiled them with (``updates``: 'adam' | 'sgd' | 'momentum' |
        'nesterov_momentum' | 'adadelta', the lasagne.updates function the script called):
            3stream/4stream: 'inputs,targets,mask,window'   (runners/3stream.py:309-320)
            2stream:         'in1,targets,mask,in2,window'  (runners/2stream.py:280-291)
            1stream:         'inputs,targets,mask,window'   (runners/1stream.py:236-247)
        ``val_fn`` takes the same order without ``targets``."""
        S = self.S
        lr = float(learning_rate)

        def split(args, with_targets):
            args = list(args)
            if order == "in1,targets,mask,in2,window":
                if with_targets:
                    in1, targets, mask, in2, window = args
                else:
                    in1, mask, in2, window = args
                    targets = None
                return [in1, in2], targets, mask, window
            ins = args[:S + len(self.aux_dims)]
            rest = args[S + len(self.aux_dims):]
            if with_targets:
                targets, mask, window = rest
            else:
                (mask, window), targets = rest, None
            return ins, targets, mask, window

        if updates not in ("adam", "sgd", "momentum", "nesterov_momentum", "adadelta"):
            raise ValueError("unknown update rule %r" % (updates,))

        def train(*args):
            ins, t, m, w = split(args, True)
            if updates == "adam":
                return self.train_step(ins, t, m, w, lr)
            cost_ = self.compute_grads(ins, t, m, w)
            if updates == "adadelta":
                self.apply_adadelta(lr, update_args.get("rho", 0.95), update_args.get("epsilon", 1e-6))
            else:
                self.apply_sgd(lr, 0.0 if updates == "sgd" else update_args.get("momentum", 0.9),
                               updates == "nesterov_momentum")
            return cost_

        def train_cost(*args):                       # get_output(network, deterministic=False): dropout active
            ins, t, m, w = split(args, True)
            return self.loss(ins, t, m, w, deterministic=False)

        def test_cost(*args):
            ins, t, m, w = split(args, True)
            return self.loss(ins, t, m, w)

        def val_fn(*args):
            ins, _, m, w = split(args, False)
            return self.predict(ins, m, w)

        return train, train_cost, test_cost, val_fn
