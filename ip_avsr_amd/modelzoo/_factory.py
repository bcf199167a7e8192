"""Shared builder behind the ``create_model`` factories of this package's model zoo.

Every graph of the reference's per-timestep-softmax family is

    S streams:  [pretrained dense encoder] -> [DeltaLayer] -> LSTM | summed BLSTM
    fusion:     none | sum | adasum | concat
    aggregation none | LSTM | summed BLSTM    ->  Dense(C)+softmax on every frame

so each zoo module only states its layer names, defaults and argument order and calls ``build``.
Initial values follow the reference: encoder W/b are the injected (DBN) arrays
(modelzoo/pretrained_encoder.py:4-9); LSTM W_in/W_hid come from ``w_init_fn``, gate biases 0
(``Gate(W_in=w_init_fn, W_hid=w_init_fn, b=Constant(0.))``, e.g. modelzoo/adenet_v2.py:20-28); peephole
vectors keep Lasagne's Gate default Normal(0.1); cell_init/hid_init 0 (learn_init=True); the softmax
DenseLayer uses Lasagne's defaults GlorotUniform / 0; adasum coefficients start at 1.0
(custom/layers.py:219-220).
"""
import numpy as np

from .. import init as _init
from ..layers import FuseHandle
from ..model import AdeNetModel, GATES

DEFAULT_ENC_NAMES = ["fc1", "fc2", "fc3", "bottleneck", "fc5", "fc6", "fc7", "fc8"]


def input_dim_of(shape):
    """(None, None, D) -> D."""
    return int(shape[-1])


def nolearn_weights(ae, shapes=(2000, 1000, 500, 50), nonlinearities=("sigmoid", "sigmoid", "sigmoid", "linear")):
    """``extract_weights`` of the last-timestep zoo modules (modelzoo/adenet_v3.py:48-61): a nolearn network object
    (``.get_all_layers()[1..4].W/.b``) -> the 4-tuple this package passes around.  A 4-tuple or a (weights, biases)
    pair passes through."""
    if isinstance(ae, (tuple, list)):
        if len(ae) == 4:
            return tuple(ae)
        weights, biases = ae
    else:
        layers = ae.get_all_layers()
        weights = [np.asarray(layers[i].W, "float32") for i in range(1, len(shapes) + 1)]
        biases = [np.asarray(layers[i].b, "float32") for i in range(1, len(shapes) + 1)]
    shapes = [int(np.asarray(w).shape[1]) for w in weights]
    return list(weights), list(biases), shapes, list(nonlinearities)[:len(shapes) - 1] + ["linear"]


def _conv_encoder_of(ae):
    """A convolutional auto-encoder offered as a stream's ``ae``: the ConvAE itself or the bottleneck handle that
    ``modelzoo.avletters_convae*.create_model`` returns beside it."""
    conv = getattr(ae, "ae", ae)
    return conv if hasattr(conv, "encode") and hasattr(conv, "bottleneck") else None


def stream(input_shape, ae=None, suffix="", delta=True, lstm_names=("lstm",), peepholes=False, enc_names=None,
           pretrained_lstm=None, pretrained_prefixes=None, dropout=0.0, batchnorm=None, aux_shape=None):
    """One stream description.  ``ae`` = (weights, biases, shapes, nonlinearities) like
    ``load_decoder`` returns (runners/3stream.py:31-40), None for an encoder-less stream, or a trained convolutional
    auto-encoder (``ip_avsr_amd.convae.ConvAE`` / its bottleneck handle): its encoder then runs as a FROZEN feature
    extractor in front of the stream (frames -> bottleneck code on the GPU, conv kernels of csrc/convae.hip) and the
    stream's delta layer / LSTM consume the code.  (The reference only ever loads the conv encoder in a dead branch,
    ``load_convae = False``, avletters/trimodal.py:278-283; training through it is outside the reference's live graph.)"""
    d = dict(input_dim=input_dim_of(input_shape), delta=bool(delta), lstm_names=list(lstm_names),
             peepholes=bool(peepholes), dropout=float(dropout), pretrained_lstm=pretrained_lstm,
             pretrained_prefixes=list(pretrained_prefixes or []), batchnorm=batchnorm,
             aux_dim=input_dim_of(aux_shape) if aux_shape is not None else 0)
    conv = _conv_encoder_of(ae)
    if conv is not None:
        if conv.D != d["input_dim"]:
            raise ValueError("the conv encoder takes %d-pixel frames, the stream declares %d" % (conv.D, d["input_dim"]))
        d.update(conv_encoder=conv, frame_dim=conv.D, input_dim=int(conv.bottleneck))
        ae = None
    if ae is None:
        d.update(enc_names=[], enc_shapes=[], enc_acts=[], enc_weights=[], enc_biases=[])
    else:
        weights, biases, shapes, nonlins = ae
        n = len(shapes)
        names = list(enc_names) if enc_names else [nm + suffix for nm in DEFAULT_ENC_NAMES[:n]]
        d.update(enc_names=names, enc_shapes=[int(s) for s in shapes], enc_acts=list(nonlins),
                 enc_weights=list(weights[:n]), enc_biases=list(biases[:n]))
    return d


def _init_lstm(model, name, w_init, peepholes):
    for g in GATES:
        for kind in ("W_in_to_", "W_hid_to_"):
            pname = "%s.%s%s" % (name, kind, g)
            model.set_param(pname, w_init(model.params[model.param_index[pname]].shape))
    if peepholes:
        peep = _init.Normal(0.1)
        for g in ("ingate", "forgetgate", "outgate"):
            pname = "%s.W_cell_to_%s" % (name, g)
            model.set_param(pname, peep(model.params[model.param_index[pname]].shape))


def _load_pretrained_lstm(model, name, weights, prefix):
    """custom/layers.py:28-52 create_pretrained_lstm: keys '{prefix}_w_{in,hid}_to_{gate}', '{prefix}_b_{gate}'."""
    for g in GATES:
        model.set_param("%s.W_in_to_%s" % (name, g), np.asarray(weights["%s_w_in_to_%s" % (prefix, g)], "float32"))
        model.set_param("%s.W_hid_to_%s" % (name, g), np.asarray(weights["%s_w_hid_to_%s" % (prefix, g)], "float32"))
        model.set_param("%s.b_%s" % (name, g), np.asarray(weights["%s_b_%s" % (prefix, g)], "float32").reshape(-1))


def lstm_param_names(name, peepholes):
    """Lasagne's registration order inside one LSTMLayer: per gate W_in, W_hid, b in the order
    ingate, forgetgate, cell, outgate; then the peephole vectors (``peepholes=True``, Lasagne's default); then
    cell_init, hid_init (``learn_init=True``)."""
    out = []
    for g in GATES:
        out += ["%s.W_in_to_%s" % (name, g), "%s.W_hid_to_%s" % (name, g), "%s.b_%s" % (name, g)]
    if peepholes:
        out += ["%s.W_cell_to_%s" % (name, g) for g in ("ingate", "forgetgate", "outgate")]
    return out + [name + ".cell_init", name + ".hid_init"]


def param_names(spec):
    """``[p.name for p in lasagne.layers.get_all_params(network)]`` of the graph ``spec`` describes -- the order
    ``get_all_param_values`` / ``set_all_param_values`` and the ``.pkl`` checkpoints use (utils/io.py:40-48).  Pure
    Python (no device): the model built from the same spec must report exactly this list."""
    names = []
    for s in spec["streams"]:
        for n in s["enc_names"]:
            names += [n + ".W", n + ".b"]
        if s.get("batchnorm"):
            names += ["%s.%s" % (s["batchnorm"], k) for k in ("beta", "gamma", "mean", "inv_std")]
        for ln in s["lstm_names"]:
            names += lstm_param_names(ln, s["peepholes"])
    if spec["fusion"] == "adasum":
        names += ["%s.adacoeff%d" % (spec["fuse_name"], k) for k in range(len(spec["streams"]))]
    for ln in spec["agg_names"]:
        names += lstm_param_names(ln, spec.get("agg_peepholes", False))
    return names + [spec["softmax_name"] + ".W", spec["softmax_name"] + ".b"]


def layer_listing(spec):
    """``[(layer.name, layer.output_shape) for layer in lasagne.layers.get_all_layers(network)]`` of the graph ``spec``
    describes, as ``utils/plotting_utils.print_network`` prints it: Lasagne lists a layer behind everything it depends on,
    walking each layer's inputs in order (an LSTMLayer's: its incoming layer, then the mask), so the streams come in the
    order the fusion layer names them and the shared ``mask`` input right ahead of the first LSTM.  The only record of that
    order in the reference is the notebook's print-out of ``adenet_v3`` (avletters/avletters_training.ipynb, pinned by
    tests/golden/adenet_v3_layers.json); ``param_names`` is this list restricted to the layers that own parameters.
    Names outside the parameterised layers follow the zoo's ``<kind><suffix>`` pattern (modelzoo/adenet_v3.py:64-188);
    a stream's suffix is read off its encoder / LSTM names, its input layer is ``input_names[k]`` of the spec."""
    out = []
    mask_done = False
    S = len(spec["streams"])
    H = spec["stream_lstm_size"]
    for k, s in enumerate(spec["streams"]):
        ref = s["enc_names"][0] if s["enc_names"] else s["lstm_names"][0]
        sfx = ref[ref.index("_"):] if "_" in ref else ""
        D = s["input_dim"]
        out.append(((spec.get("input_names") or [])[k] if k < len(spec.get("input_names") or []) else "input" + sfx, (None, None, D)))
        width = D
        if s["enc_names"]:
            out.append(("reshape1" + sfx, (None, D)))
            for n, u in zip(s["enc_names"], s["enc_shapes"]):
                out.append((n, (None, int(u))))
            width = int(s["enc_shapes"][-1])
            out.append(("reshape2" + sfx, (None, None, width)))
        if s["delta"]:
            width *= 3
            out.append(("delta" + sfx, (None, None, width)))
        if s["dropout"] > 0:
            out.append(("dropout" + sfx, (None, None, width)))
        if not mask_done:
            out.append(("mask", (None, None)))
            mask_done = True
        for ln in s["lstm_names"]:
            out.append((ln, (None, None, H)))
    width = H
    if spec["fusion"] != "none" and S > 1:
        width = H * S if spec["fusion"] == "concat" else H
        out.append((spec["fuse_name"], (None, None, width)))
    if spec.get("agg_dropout", 0.0) > 0:
        out.append(("dropout_agg", (None, None, width)))
    for ln in spec["agg_names"]:
        out.append((ln, (None, None, spec["lstm_size"])))
        width = spec["lstm_size"]
    if len(spec["agg_names"]) == 2:
        out.append(("sum2", (None, None, width)))
    if spec.get("head") == "last":
        out.append(("slice1", (None, width)))
        out.append((spec["softmax_name"], (None, spec["classes"])))
    else:
        out.append(("reshape3", (None, width)))
        out.append((spec["softmax_name"], (None, spec["classes"])))
        out.append(("output", (None, None, spec["classes"])))
    return out


# set by tests that only need the graph description of a zoo module (no device, no libadenet_hip.so)
SPEC_ONLY = False

# arithmetic of the models the factories build ('f32' | 'bf16x3' | 'mixed' | 'bf16', include/adenet.h adn_precision).  The reference is
# fp32 throughout (floatX = float32); the drivers' ``--precision`` option / the ADN_PRECISION environment variable set this
# before they call ``create_model`` (whose reference signature has no room for it).
# The product default (round 6) is bf16x3: fp32-GRADE products on the bf16 matrix pipe -- meets the 1e-4 / exact-top-1 parity gate
# against the fp32 reference (tests/test_gpu_bf16x3.py) on the weight-stationary LSTM kernels, 5.6x the throughput of 'f32', which is
# the exact-product DIAGNOSTIC arithmetic (one launch per LSTM time step; what the exact-product parity tests ask for explicitly).
import os as _os
PRODUCT_DEFAULT_PRECISION = "bf16x3"
DEFAULT_PRECISION = _os.environ.get("ADN_PRECISION", PRODUCT_DEFAULT_PRECISION)


def set_default_precision(precision):
    global DEFAULT_PRECISION
    if precision not in ("f32", "bf16x3", "mixed", "bf16"):
        raise ValueError("precision must be f32, bf16x3, mixed or bf16 (got %r)" % (precision,))
    DEFAULT_PRECISION = precision


def build(streams, lstm_size, output_classes, fusiontype, fuse_names, agg_names, agg_peepholes, w_init_fn,
          softmax_name="softmax", return_fuse=True, head="frames", agg_dropout=0.0, stream_lstm_size=None, input_names=None):
    if fusiontype not in ("none", "sum", "adasum", "concat"):
        # modelzoo/adenet_v2.py:74-75 (other factories fall through to a NameError)
        raise ValueError("Unsupported Fusion Type used!")
    spec = dict(
        streams=[{k: s[k] for k in ("input_dim", "enc_names", "enc_shapes", "enc_acts", "delta", "lstm_names",
                                    "peepholes", "dropout", "batchnorm", "aux_dim")} for s in streams],
        head=head, agg_dropout=float(agg_dropout), stream_lstm_size=int(stream_lstm_size or lstm_size),
        fusion=fusiontype, fuse_name=fuse_names.get(fusiontype, ""), agg_names=list(agg_names),
        agg_peepholes=bool(agg_peepholes), lstm_size=int(lstm_size), classes=int(output_classes),
        softmax_name=softmax_name, precision=DEFAULT_PRECISION)
    if input_names:
        spec["input_names"] = list(input_names)        # (layer names of the InputLayers: layer_listing only)
    if SPEC_ONLY:
        return (spec, None) if return_fuse else spec
    model = AdeNetModel(spec)
    assert [p.name for p in model.params] == param_names(spec)
    for k, s in enumerate(streams):
        if s.get("conv_encoder") is not None:
            model.set_front_end(k, s["conv_encoder"], s["frame_dim"])
    w_init = _init.resolve(w_init_fn)
    for s in streams:
        for n, W, b in zip(s["enc_names"], s["enc_weights"], s["enc_biases"]):
            model.set_param(n + ".W", W)
            model.set_param(n + ".b", np.asarray(b).reshape(-1))
        if s.get("batchnorm"):                       # lasagne.layers.BatchNormLayer: beta 0, gamma 1, mean 0, inv_std 1
            n = model.params[model.param_index[s["batchnorm"] + ".gamma"]].shape
            model.set_param(s["batchnorm"] + ".gamma", np.ones(n, "float32"))
            model.set_param(s["batchnorm"] + ".inv_std", np.ones(n, "float32"))
        for k, ln in enumerate(s["lstm_names"]):
            _init_lstm(model, ln, w_init, s["peepholes"])
            if s["pretrained_lstm"] is not None:
                _load_pretrained_lstm(model, ln, s["pretrained_lstm"], s["pretrained_prefixes"][k])
    if fusiontype == "adasum":
        for k in range(len(streams)):
            model.set_param("%s.adacoeff%d" % (spec["fuse_name"], k), np.float32(1.0))
    for ln in agg_names:
        _init_lstm(model, ln, w_init, agg_peepholes)
    sm_shape = model.params[model.param_index[softmax_name + ".W"]].shape
    model.set_param(softmax_name + ".W", _init.GlorotUniform()(sm_shape))
    if return_fuse:
        return model, FuseHandle(model, spec["fuse_name"])
    return model
