"""adenet_v3's graph with a switch between ElemwiseSum 'sum1' and AdaptiveElemwiseSum 'adasum1' for the fusion:
raw + diff encoder streams with deltas (dropout 0.5), a DCT stream (dropout 0.2), LSTMs of ``lstm_size / (1 - 0.5)``
units, dropout 0.5 on the fused tensor, summed BLSTM 'f_lstm_agg' / 'b_lstm_agg' of ``2 * lstm_size`` units,
``SliceLayer(-1)``, softmax 'output' (reference modelzoo/adenet_v5.py:64-186)."""
from . import _factory as F


def create_model(ae, diff_ae, input_shape, input_var, mask_shape, mask_var, dct_shape, dct_var, diff_shape, diff_var,
                 lstm_size=250, win=None, output_classes=26, use_adascale=False):
    wide = int(lstm_size / (1 - 0.5))
    streams = [F.stream(input_shape, F.nolearn_weights(ae), "_raw", lstm_names=["lstm_raw"], dropout=0.5, peepholes=True),
               F.stream(dct_shape, None, "_dct", delta=False, lstm_names=["lstm_dct"], dropout=0.2, peepholes=True),
               F.stream(diff_shape, F.nolearn_weights(diff_ae), "_diff", lstm_names=["lstm_diff"], dropout=0.5, peepholes=True)]
    return F.build(streams, wide, output_classes, "adasum" if use_adascale else "sum", {"sum": "sum1", "adasum": "adasum1"},
                   ["f_lstm_agg", "b_lstm_agg"], True, 'ortho', softmax_name="output", head="last", agg_dropout=0.5)
