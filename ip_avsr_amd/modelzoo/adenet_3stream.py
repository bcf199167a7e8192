"""3-stream AdeNet: three encoder streams (reference modelzoo/adenet_3stream.py:12-142 create_pretrained_model,
:145-264 create_model) -- the graph runners/3stream.py builds."""
from . import _factory as F

_FUSE = {"sum": "sum1", "adasum": "adasum1", "concat": "concat"}


def create_model(s1_ae, s2_ae, s3_ae, s1_shape, s1_var, s2_shape, s2_var, s3_shape, s3_var, mask_shape, mask_var,
                 lstm_size=250, win=None, output_classes=26, fusiontype='concat', w_init_fn='ortho',
                 use_peepholes=True):
    streams = [F.stream(shp, ae, "_s%d" % (k + 1), lstm_names=["lstm_s%d" % (k + 1)], peepholes=use_peepholes)
               for k, (shp, ae) in enumerate(((s1_shape, s1_ae), (s2_shape, s2_ae), (s3_shape, s3_ae)))]
    return F.build(streams, lstm_size, output_classes, fusiontype, _FUSE, ["f_lstm_agg", "b_lstm_agg"], False,
                   w_init_fn)


def create_pretrained_model(s1_ae, s1_lstm, s2_ae, s2_lstm, s3_ae, s3_lstm, s1_shape, s1_var, s2_shape, s2_var,
                            s3_shape, s3_var, mask_shape, mask_var, lstm_size=250, win=None, output_classes=26,
                            fusiontype='concat', w_init_fn='ortho', use_peepholes=True, use_blstm_substream=False):
    streams = []
    for k, (shp, ae, lw) in enumerate(((s1_shape, s1_ae, s1_lstm), (s2_shape, s2_ae, s2_lstm),
                                       (s3_shape, s3_ae, s3_lstm)), 1):
        if use_blstm_substream:
            names, prefixes = ["f_lstm_s%d" % k, "b_lstm_s%d" % k], ["f_lstm", "b_lstm"]
        else:
            names, prefixes = ["f_lstm_s%d" % k], ["f_lstm"]
        streams.append(F.stream(shp, ae, "_s%d" % k, lstm_names=names, peepholes=use_peepholes, pretrained_lstm=lw,
                                pretrained_prefixes=prefixes))
    return F.build(streams, lstm_size, output_classes, fusiontype, _FUSE, ["f_lstm_agg", "b_lstm_agg"], False,
                   w_init_fn)
