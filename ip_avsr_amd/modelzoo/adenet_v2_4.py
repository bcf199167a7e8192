"""AdeNet v2.4: raw + diff-image encoder streams, single forward aggregation LSTM
(reference modelzoo/adenet_v2_4.py:12-29,32-123)."""
from . import _factory as F


def create_model(ae, diff_ae, input_shape, input_var, mask_shape, mask_var, diff_shape, diff_var, lstm_size=250,
                 win=None, output_classes=26, fusiontype='concat', w_init_fn='ortho', use_peepholes=True):
    streams = [F.stream(input_shape, ae, "_raw", lstm_names=["lstm_raw"], peepholes=use_peepholes),
               F.stream(diff_shape, diff_ae, "_diff", lstm_names=["lstm_diff"], peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, fusiontype,
                   {"sum": "sum1", "adasum": "adasum1", "concat": "concat"},
                   ["f_lstm_agg"], True, w_init_fn)
