"""DeltaNet v1: delta layer directly on the input features, (B)LSTM, per-frame softmax
(reference modelzoo/deltanet_v1.py:8-42)."""
from . import _factory as F


def create_model(input_shape, input_var, mask_shape, mask_var, window, lstm_size=250, output_classes=26,
                 w_init='glorot', use_peepholes=False, use_blstm=True):
    names = ["f_lstm", "b_lstm"] if use_blstm else ["lstm"]
    streams = [F.stream(input_shape, None, delta=True, lstm_names=names, peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, "none", {}, [], False, w_init, return_fuse=False)
