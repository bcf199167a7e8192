"""AdeNet v2.2: two encoder streams; the module's own create_blstm defaults to peepholes in the
aggregation BLSTM (reference modelzoo/adenet_v2_2.py:12-37,40-132)."""
from . import _factory as F


def create_model(ae, s2_ae, input_shape, input_var, mask_shape, mask_var, s2_shape, s2_var, lstm_size=250, win=None,
                 output_classes=26, fusiontype='concat', w_init_fn='ortho', use_peepholes=True):
    streams = [F.stream(input_shape, ae, "_s1", lstm_names=["lstm_s1"], peepholes=use_peepholes),
               F.stream(s2_shape, s2_ae, "_s2", lstm_names=["lstm_s2"], peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, fusiontype,
                   {"sum": "sum1", "adasum": "adasum1", "concat": "concat"},
                   ["f_lstm_agg", "b_lstm_agg"], True, w_init_fn)
