"""AdeNet v1 (reference modelzoo/adenet_v1.py:48-109): encoder 'fc1' .. 'bottleneck' (sigmoid-sigmoid-sigmoid-linear,
:11-16) -> BatchNormLayer 'batchnorm1' on the (B*T, 50) codes -> DeltaLayer -> ConcatLayer with the DCT input (axis 2,
'concat') -> summed BLSTM 'f_lstm1' / 'b_lstm1' of ``lstm_size`` units -> summed BLSTM 'f_lstm2' / 'b_lstm2' of
``2 * lstm_size`` units -> ``SliceLayer(-1)`` -> softmax 'output'; returns (network, the concat layer).
No LSTMLayer passes ``peepholes=``: Lasagne's default True applies (:27-43).  The narrower first BLSTM runs inside
``2 * lstm_size``-wide kernels with its surplus units pinned at zero (include/adenet.h: stream_lstm_units); parameter
shapes and the ``get_all_param_values`` order are the reference's."""
from . import _factory as F


def create_model(dbn, input_shape, input_var, mask_shape, mask_var, dct_shape, dct_var, lstm_size=250, win=None,
                 output_classes=26):
    streams = [F.stream(input_shape, F.nolearn_weights(dbn), "", lstm_names=["f_lstm1", "b_lstm1"], peepholes=True,
                        batchnorm="batchnorm1", aux_shape=dct_shape)]
    return F.build(streams, 2 * int(lstm_size), output_classes, "none", {"none": "concat"}, ["f_lstm2", "b_lstm2"], True, 'ortho',
                   softmax_name="output", head="last", stream_lstm_size=int(lstm_size))
