"""AdeNet v2.3: v2 with a single forward aggregation LSTM (named f_lstm_agg, peepholes on by the
module's own create_lstm default) (reference modelzoo/adenet_v2_3.py:41-58,61-149)."""
from . import _factory as F


def create_model(dbn, input_shape, input_var, mask_shape, mask_var, dct_shape, dct_var, lstm_size=250, win=None,
                 output_classes=26, fusiontype='sum', w_init_fn='ortho', use_peepholes=True):
    streams = [F.stream(input_shape, dbn, delta=True, lstm_names=["lstm_bn"], peepholes=use_peepholes),
               F.stream(dct_shape, None, delta=False, lstm_names=["lstm_dct"], peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, fusiontype,
                   {"sum": "sum1", "adasum": "adasum", "concat": "concat"},
                   ["f_lstm_agg"], True, w_init_fn)
