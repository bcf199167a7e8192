"""Three encoder streams with deltas, dropout 0.5 ahead of every stream LSTM and on the fused tensor, every LSTM of
``2 * lstm_size`` units, per-frame softmax (reference modelzoo/adenet_3stream_dropout.py:13-139)."""
from . import _factory as F


def create_model(s1_ae, s2_ae, s3_ae, s1_shape, s1_var, s2_shape, s2_var, s3_shape, s3_var, mask_shape, mask_var,
                 lstm_size=250, win=None, output_classes=26, fusiontype='concat', w_init_fn='ortho', use_peepholes=True):
    streams = [F.stream(s1_shape, s1_ae, "_s1", lstm_names=["lstm_s1"], peepholes=use_peepholes, dropout=0.5),
               F.stream(s2_shape, s2_ae, "_s2", lstm_names=["lstm_s2"], peepholes=use_peepholes, dropout=0.5),
               F.stream(s3_shape, s3_ae, "_s3", lstm_names=["lstm_s3"], peepholes=use_peepholes, dropout=0.5)]
    return F.build(streams, lstm_size * 2, output_classes, fusiontype, {"sum": "sum1", "adasum": "adasum1", "concat": "concat"},
                   ["f_lstm_agg", "b_lstm_agg"], False, w_init_fn, agg_dropout=0.5)
