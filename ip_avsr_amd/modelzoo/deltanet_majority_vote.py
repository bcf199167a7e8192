"""Single-stream DeltaNet with a per-frame softmax (majority-vote evaluation)
(reference modelzoo/deltanet_majority_vote.py:14-66 create_model, :69-134 load_saved_model,
:137-196 extract_*)."""
import numpy as np

from . import _factory as F
from ..model import GATES
from ..utils.io import load_model_params


def create_model(dbn, input_shape, input_var, mask_shape, mask_var, lstm_size=250, win=None, output_classes=26,
                 w_init_fn='glorot', use_peepholes=False, use_blstm=True):
    names = ["f_blstm1", "b_blstm1"] if use_blstm else ["lstm"]
    streams = [F.stream(input_shape, dbn, "", lstm_names=names, peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, "none", {}, [], False, w_init_fn, return_fuse=False)


def create_model_using_pretrained_encoder(weights, biases, input_shape, input_var, mask_shape, mask_var, lstm_size=250,
                                          win=None, output_classes=26, w_init_fn='ortho', use_peepholes=False,
                                          nonlinearities='rectify'):
    """What cuave/unimodal_with_val.py:259-263 and oulu/unimodal_with_val.py:311-314 call.  The reference's
    modelzoo/deltanet_majority_vote.py does NOT define it (those two scripts fail with AttributeError as shipped); the only
    function of this name, modelzoo/deltanet.py:12-56, has this argument list but the last-timestep head, while both scripts
    train with temporal_softmax_loss on per-frame targets and evaluate by majority vote.  So: deltanet.py's argument list
    and encoder construction (``[nonlinearities] * 3 + [linear]``, layer names fc1..bottleneck), this module's per-frame
    summed-BLSTM graph."""
    n = len(weights)
    ae = (list(weights), list(biases), [int(w.shape[1]) for w in weights], [nonlinearities] * (n - 1) + ["linear"])
    return create_model(ae, input_shape, input_var, mask_shape, mask_var, lstm_size, win, output_classes, w_init_fn,
                        use_peepholes, use_blstm=True)


def load_saved_model(model_path, stream_params, input_shape, input_var, mask_shape, mask_var, lstm_size=250, win=None,
                     output_classes=26, w_init_fn='glorot', use_peepholes=False, use_blstm=True):
    """Rebuild the graph with a fresh (randomly initialised) encoder of the given widths and load a
    pickled parameter list into it.  ``stream_params`` = (shapes, nonlinearities)."""
    shapes, nonlins = stream_params
    d = F.input_dim_of(input_shape)
    weights, biases = [], []
    from .. import init as _init
    for u in shapes:
        weights.append(_init.GlorotUniform()((d, int(u))))
        biases.append(np.zeros((int(u),), "float32"))
        d = int(u)
    net = create_model((weights, biases, list(shapes), list(nonlins)), input_shape, input_var, mask_shape, mask_var,
                       lstm_size, win, output_classes, w_init_fn, use_peepholes, use_blstm)
    return load_model_params(net, model_path)


def extract_encoder_weights(network, names, params):
    """dict w1..wN / b1..bN of the named encoder layers (keys as written to .mat by the reference)."""
    out = {}
    for layer, (wk, bk) in zip(names, params):
        out[wk] = network.get_param(layer + ".W")
        out[bk] = network.get_param(layer + ".b")
    return out


def extract_lstm_weights(network, names, prefixes):
    """dict '{prefix}_w_{in,hid}_to_{gate}', '{prefix}_b_{gate}' for each named LSTM layer
    (reference :158-196; the format create_pretrained_lstm consumes)."""
    out = {}
    for layer, prefix in zip(names, prefixes):
        for g in GATES:
            out["%s_w_in_to_%s" % (prefix, g)] = network.get_param("%s.W_in_to_%s" % (layer, g))
            out["%s_w_hid_to_%s" % (prefix, g)] = network.get_param("%s.W_hid_to_%s" % (layer, g))
            out["%s_b_%s" % (prefix, g)] = network.get_param("%s.b_%s" % (layer, g))
    return out
