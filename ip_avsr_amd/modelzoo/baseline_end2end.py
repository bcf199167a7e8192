"""Encoder (sigmoid-sigmoid-sigmoid-linear, 'fc1' .. 'bottleneck') -> summed BLSTM 'f_lstm1' / 'b_lstm1' on the
bottleneck features WITHOUT deltas -> ``SliceLayer(-1)`` -> softmax 'output' (reference modelzoo/baseline_end2end.py:64-116)."""
from . import _factory as F


def create_model(dbn, input_shape, input_var, mask_shape, mask_var, lstm_size=250, output_classes=26):
    streams = [F.stream(input_shape, F.nolearn_weights(dbn), "", delta=False, lstm_names=["f_lstm1", "b_lstm1"], peepholes=True)]
    return F.build(streams, lstm_size, output_classes, "none", {}, [], False, 'ortho', softmax_name="output",
                   return_fuse=False, head="last")
