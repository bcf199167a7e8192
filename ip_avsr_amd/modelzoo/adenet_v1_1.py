"""AdeNet v1.1 (reference modelzoo/adenet_v1_1.py:48-114): adenet_v1 with DropoutLayer 'dropout1' (p = 0.5) on the
[deltas | DCT] concatenation, BOTH summed BLSTMs ('f_lstm1' / 'b_lstm1', 'f_lstm2' / 'b_lstm2') of ``2 * lstm_size``
units and DropoutLayer 'dropout2' between them; returns the network only (:114)."""
from . import _factory as F


def create_model(dbn, input_shape, input_var, mask_shape, mask_var, dct_shape, dct_var, lstm_size=250, win=None,
                 output_classes=26):
    streams = [F.stream(input_shape, F.nolearn_weights(dbn), "", lstm_names=["f_lstm1", "b_lstm1"], peepholes=True,
                        batchnorm="batchnorm1", aux_shape=dct_shape, dropout=0.5)]
    return F.build(streams, 2 * int(lstm_size), output_classes, "none", {"none": "concat"}, ["f_lstm2", "b_lstm2"], True, 'ortho',
                   softmax_name="output", head="last", agg_dropout=0.5, return_fuse=False)
