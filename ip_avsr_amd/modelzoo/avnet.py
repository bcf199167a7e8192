"""AVNet: sub-streams are created one by one, then fused (reference modelzoo/avnet.py:30-71
create_pretrained_substream, :74-114 create_model).  A sub-stream here is a deferred description; the graph
is materialised by ``create_model``."""
from . import _factory as F


def create_pretrained_substream(weights, biases, input_shape, input_var, mask_shape, mask_var, name, lstm_size=250,
                                win=None, nonlinearity='rectify', w_init_fn='ortho', use_peepholes=True):
    n = len(weights[:4])
    acts = [nonlinearity] * (n - 1) + ['linear']               # avnet.py:47-48
    shapes = [int(w.shape[1]) for w in weights[:n]]
    names = ["%s_%s" % (nm, name) for nm in F.DEFAULT_ENC_NAMES[:n]]
    return F.stream(input_shape, (list(weights[:n]), list(biases[:n]), shapes, acts), enc_names=names,
                    lstm_names=["lstm_" + name], peepholes=use_peepholes)


def create_model(substreams, mask_shape, mask_var, lstm_size=250, output_classes=26, fusiontype='concat',
                 w_init_fn='ortho', use_peepholes=True):
    return F.build(list(substreams), lstm_size, output_classes, fusiontype,
                   {"sum": "sum1", "adasum": "adasum1", "concat": "concat"},
                   ["f_lstm_agg", "b_lstm_agg"], False, w_init_fn)
