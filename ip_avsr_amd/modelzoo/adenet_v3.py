"""Last-timestep AdeNet: raw + diff encoder streams with deltas and a raw DCT stream, dropout 0.5 / 0.2 / 0.5 ahead of
the three LSTMs of ``lstm_size / (1 - 0.5)`` units, dropout 0.5 on the fused tensor, a summed BLSTM of ``2 * lstm_size``
units, ``SliceLayer(-1)`` and a softmax layer named 'output'; trained with categorical cross-entropy
(reference modelzoo/adenet_v3.py:64-188, avletters/trimodal.py:327-328).  ``ae`` / ``diff_ae``: nolearn networks, or
the (weights, biases[, shapes, nonlinearities]) tuples; the encoders are sigmoid-sigmoid-sigmoid-linear (:12-17)."""
from . import _factory as F


def create_model(ae, diff_ae, input_shape, input_var, mask_shape, mask_var, dct_shape, dct_var, diff_shape, diff_var,
                 lstm_size=250, win=None, output_classes=26, fusiontype='sum'):
    wide = int(lstm_size / (1 - 0.5))
    streams = [F.stream(input_shape, F.nolearn_weights(ae), "_raw", lstm_names=["lstm_raw"], dropout=0.5, peepholes=True),
               F.stream(dct_shape, None, "_dct", delta=False, lstm_names=["lstm_dct"], dropout=0.2, peepholes=True),
               F.stream(diff_shape, F.nolearn_weights(diff_ae), "_diff", lstm_names=["lstm_diff"], dropout=0.5, peepholes=True)]
    return F.build(streams, wide, output_classes, fusiontype, {"sum": "sum1", "adasum": "adasum1", "concat": "concat"},
                   ["f_lstm_agg", "b_lstm_agg"], True, 'ortho', softmax_name="output", head="last", agg_dropout=0.5,
                   input_names=["raw_im", "dct", "diff_im"])
