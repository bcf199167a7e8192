"""(B)LSTM classifier on raw features, per-frame softmax (reference modelzoo/lstm_classifier_majority_vote.py:10-43)."""
from . import _factory as F


def create_model(input_shape, input_var, mask_shape, mask_var, lstm_size=250, output_classes=26, w_init='glorot',
                 use_peepholes=False, use_blstm=True):
    names = ["f_lstm", "b_lstm"] if use_blstm else ["lstm"]
    streams = [F.stream(input_shape, None, delta=False, lstm_names=names, peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, "none", {}, [], False, w_init, return_fuse=False)
