"""Last-timestep AdeNet with a raw encoder stream (deltas, dropout 0.5) and a DCT stream (dropout 0.2), LSTMs of
``2 * lstm_size`` units, ElemwiseSum 'sum1', dropout 0.5 on the sum, ONE forward aggregation LSTM 'lstm_agg' of
``2 * lstm_size`` units, ``SliceLayer(-1)`` and the softmax layer 'output' (reference modelzoo/adenet_v4.py:48-147).
``dbn``: a nolearn network or the (weights, biases[, shapes, nonlinearities]) tuple; sigmoid-sigmoid-sigmoid-linear
encoder 'fc1' .. 'bottleneck' (:11-17)."""
from . import _factory as F


def create_model(dbn, input_shape, input_var, mask_shape, mask_var, dct_shape, dct_var, lstm_size=250, win=None,
                 output_classes=26):
    streams = [F.stream(input_shape, F.nolearn_weights(dbn), "", lstm_names=["lstm_bn"], dropout=0.5, peepholes=True),
               F.stream(dct_shape, None, "_dct", delta=False, lstm_names=["lstm_dct"], dropout=0.2, peepholes=True)]
    return F.build(streams, 2 * int(lstm_size), output_classes, "sum", {"sum": "sum1"}, ["lstm_agg"], True, 'ortho',
                   softmax_name="output", head="last", agg_dropout=0.5)
