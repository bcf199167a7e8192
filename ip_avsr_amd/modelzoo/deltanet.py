"""Last-timestep DeltaNet: encoder -> deltas -> summed BLSTM -> ``SliceLayer(-1)`` -> softmax 'output'
(reference modelzoo/deltanet.py:12-77)."""
from . import _factory as F


def create_model_using_pretrained_encoder(weights, biases, input_shape, input_var, mask_shape, mask_var, lstm_size=250,
                                          win=None, output_classes=26, w_init_fn='ortho', use_peepholes=False,
                                          nonlinearities='rectify'):
    n = len(weights)
    ae = (list(weights), list(biases), [int(w.shape[1]) for w in weights], [nonlinearities] * (n - 1) + ["linear"])
    streams = [F.stream(input_shape, ae, "", lstm_names=["f_bstm1", "b_bstm1"], peepholes=use_peepholes)]
    return F.build(streams, lstm_size, output_classes, "none", {}, [], False, w_init_fn, softmax_name="output",
                   return_fuse=False, head="last")


def create_model(dbn, input_shape, input_var, mask_shape, mask_var, lstm_size=250, win=None, output_classes=26):
    weights, biases, _, _ = F.nolearn_weights(dbn)
    return create_model_using_pretrained_encoder(weights, biases, input_shape, input_var, mask_shape, mask_var, lstm_size,
                                                 win, output_classes)
