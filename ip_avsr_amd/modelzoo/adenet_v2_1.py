"""Last-timestep bimodal AdeNet: raw + diff encoder streams with deltas, LSTMs of ``lstm_size`` units, fusion, a
summed BLSTM, ``SliceLayer(-1)`` and softmax 'output' (reference modelzoo/adenet_v2_1.py:58-175; its dropout layers are
commented out in the live graph)."""
from . import _factory as F


def create_model(ae, diff_ae, input_shape, input_var, mask_shape, mask_var, diff_shape, diff_var, lstm_size=250,
                 win=None, output_classes=26, fusiontype='concat', w_init_fn='ortho', use_peepholes=True):
    relu = ("rectify", "rectify", "rectify", "linear")
    streams = [F.stream(input_shape, F.nolearn_weights(ae, nonlinearities=relu), "_raw", lstm_names=["lstm_raw"],
                        peepholes=use_peepholes),
               F.stream(diff_shape, F.nolearn_weights(diff_ae, nonlinearities=relu), "_diff", lstm_names=["lstm_diff"],
                        peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, fusiontype, {"sum": "sum1", "adasum": "adasum1", "concat": "concat"},
                   ["f_lstm_agg", "b_lstm_agg"], use_peepholes, w_init_fn, softmax_name="output", head="last")
