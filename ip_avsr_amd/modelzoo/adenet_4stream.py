"""4-stream AdeNet (reference modelzoo/adenet_4stream.py:12-159) -- the graph runners/4stream.py builds."""
from . import _factory as F


def create_model(s1_ae, s2_ae, s3_ae, s4_ae, s1_shape, s1_var, s2_shape, s2_var, s3_shape, s3_var, s4_shape, s4_var,
                 mask_shape, mask_var, lstm_size=250, win=None, output_classes=26, fusiontype='concat',
                 w_init_fn='ortho', use_peepholes=True):
    pairs = ((s1_shape, s1_ae), (s2_shape, s2_ae), (s3_shape, s3_ae), (s4_shape, s4_ae))
    streams = [F.stream(shp, ae, "_s%d" % (k + 1), lstm_names=["lstm_s%d" % (k + 1)], peepholes=use_peepholes)
               for k, (shp, ae) in enumerate(pairs)]
    return F.build(streams, lstm_size, output_classes, fusiontype,
                   {"sum": "sum1", "adasum": "adasum1", "concat": "concat"},
                   ["f_lstm_agg", "b_lstm_agg"], False, w_init_fn)
