"""adenet_v5 without the DCT stream: raw + diff encoder streams with deltas and dropout 0.5, LSTMs of
``lstm_size / (1 - 0.5)`` units, sum or adaptive sum, dropout 0.5, summed BLSTM of ``2 * lstm_size`` units, last time
step, softmax 'output' (reference modelzoo/adenet_v6.py:64-177)."""
from . import _factory as F


def create_model(ae, diff_ae, input_shape, input_var, mask_shape, mask_var, diff_shape, diff_var, lstm_size=250, win=None,
                 output_classes=26, use_adascale=False):
    wide = int(lstm_size / (1 - 0.5))
    streams = [F.stream(input_shape, F.nolearn_weights(ae), "_raw", lstm_names=["lstm_raw"], dropout=0.5, peepholes=True),
               F.stream(diff_shape, F.nolearn_weights(diff_ae), "_diff", lstm_names=["lstm_diff"], dropout=0.5, peepholes=True)]
    return F.build(streams, wide, output_classes, "adasum" if use_adascale else "sum", {"sum": "sum1", "adasum": "adasum1"},
                   ["f_lstm_agg", "b_lstm_agg"], True, 'ortho', softmax_name="output", head="last", agg_dropout=0.5)
