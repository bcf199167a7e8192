"""INI string -> nonlinearity name (reference custom/nonlinearities.py:4-16).  The HIP path implements the
ones the DBN encoders use; the rest raise when a model is built."""

SUPPORTED = ("rectify", "sigmoid", "leaky_rectify", "very_leaky_rectify", "tanh", "linear", "identity")


def select_nonlinearity(string):
    table = {k: k for k in SUPPORTED}
    table.update({k: k for k in ("softmax", "softplus", "elu", "scaled_tanh")})   # accepted names, unsupported on device
    return table[string]
