"""AdeNet v2: one encoder stream + one raw (DCT) stream, each delta'd, two LSTMs, fusion, summed BLSTM,
per-frame softmax (reference modelzoo/adenet_v2.py:12-94)."""
from . import _factory as F


def create_model(dbn, input_shape, input_var, mask_shape, mask_var, dct_shape, dct_var, lstm_size=250, win=None,
                 output_classes=26, fusiontype='sum', w_init_fn='glorot', use_peepholes=False, nonlinearities=None):
    # ``nonlinearities`` is shadowed by the dbn tuple's entry in the reference (adenet_v2.py:17) and ignored
    streams = [F.stream(input_shape, dbn, delta=True, lstm_names=["lstm_bn"], peepholes=use_peepholes),
               F.stream(dct_shape, None, delta=True, lstm_names=["lstm_dct"], peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, fusiontype,
                   {"sum": "sum1", "adasum": "adasum", "concat": "concat"},
                   ["f_lstm_agg", "b_lstm_agg"], False, w_init_fn)
