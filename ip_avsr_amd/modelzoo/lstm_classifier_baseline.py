"""Summed BLSTM 'f_lstm' / 'b_lstm' on the raw features, ``SliceLayer(-1)``, softmax 'output'
(reference modelzoo/lstm_classifier_baseline.py:56-82)."""
from . import _factory as F


def create_model(input_shape, input_var, mask_shape, mask_var, lstm_size=250, output_classes=26, w_init='ortho'):
    streams = [F.stream(input_shape, None, delta=False, lstm_names=["f_lstm", "b_lstm"], peepholes=True)]
    return F.build(streams, lstm_size, output_classes, "none", {}, [], False, w_init, softmax_name="output",
                   return_fuse=False, head="last")
