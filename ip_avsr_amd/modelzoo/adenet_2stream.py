"""2-stream AdeNet (reference modelzoo/adenet_2stream.py:12-113 create_pretrained_model, :116-210 create_model)."""
from . import _factory as F

_FUSE = {"sum": "sum1", "adasum": "adasum1", "concat": "concat"}


def create_model(s1_ae, s2_ae, s1_shape, s1_var, s2_shape, s2_var, mask_shape, mask_var, lstm_size=250, win=None,
                 output_classes=26, fusiontype='concat', w_init_fn='ortho', use_peepholes=True):
    streams = [F.stream(s1_shape, s1_ae, "_s1", lstm_names=["lstm_s1"], peepholes=use_peepholes),
               F.stream(s2_shape, s2_ae, "_s2", lstm_names=["lstm_s2"], peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, fusiontype, _FUSE, ["f_lstm_agg", "b_lstm_agg"], False,
                   w_init_fn)


def create_pretrained_model(s1_ae, s1_lstm, s2_ae, s2_lstm, s1_shape, s1_var, s2_shape, s2_var, mask_shape, mask_var,
                            lstm_size=250, win=None, output_classes=26, fusiontype='concat', w_init_fn='ortho',
                            use_peepholes=True, use_blstm_substream=False):
    """Sub-stream LSTMs start from extracted single-stream weights ({f,b}_lstm_* keys,
    custom/layers.py:28-52); with ``use_blstm_substream`` each stream is a summed f/b pair."""
    def names(k):
        return (["f_lstm_s%d" % k, "b_lstm_s%d" % k], ["f_lstm", "b_lstm"]) if use_blstm_substream \
            else (["f_lstm_s%d" % k], ["f_lstm"])
    n1, p1 = names(1)
    n2, p2 = names(2)
    streams = [F.stream(s1_shape, s1_ae, "_s1", lstm_names=n1, peepholes=use_peepholes, pretrained_lstm=s1_lstm,
                        pretrained_prefixes=p1),
               F.stream(s2_shape, s2_ae, "_s2", lstm_names=n2, peepholes=use_peepholes, pretrained_lstm=s2_lstm,
                        pretrained_prefixes=p2)]
    return F.build(streams, lstm_size, output_classes, fusiontype, _FUSE, ["f_lstm_agg", "b_lstm_agg"], False,
                   w_init_fn)
