"""Optimiser entry points with the reference's names.

``adam`` = lasagne.updates.adam as the runners call it (runners/3stream.py:307); ``generate_lr_map`` / ``adam_vlr`` =
reference custom/updates.py:10-32 / :35-99 (Adam with a learning rate per layer, used by
runners/1stream_variable_lr.py:235-243,327-333).  The arithmetic is csrc/elementwise.hip::adam_kernel; here the
"update" objects are callables bound to a model instead of Theano update dictionaries."""

BETA1, BETA2, EPSILON = 0.9, 0.999, 1e-8


def adam(model, learning_rate=1e-3):
    """Returns the update callable for ``model`` (one Adam step on its current gradient buffer)."""
    def step():
        model.apply_adam(learning_rate)
    return step


def generate_lr_map(params, lr_config, default):
    """Per-parameter learning rates from a per-LAYER configuration: the layer name is the parameter name up to
    its last dot (e.g. 'fc1.W' -> 'fc1'); layers missing from ``lr_config`` get ``default``."""
    lr_map = {}
    for param in params:
        layer_name = param.name[:param.name.rfind('.')]
        lr_map[param] = lr_config.get(layer_name, default)
    return lr_map


def adam_vlr(model, lr_map):
    """Adam with variable learning rates (one shared step counter, like the reference)."""
    def step():
        model.apply_adam_vlr(lr_map)
    return step


# lasagne.updates.* rules the last-timestep scripts select (avletters/bimodal.py:446-455, avletters/trimodal.py:330-336,
# avletters/avletters_convae.py:230): same names and defaults, bound to a model like ``adam`` above
def sgd(model, learning_rate):
    def step():
        model.apply_sgd(learning_rate)
    return step


def momentum(model, learning_rate, momentum=0.9):
    mu = momentum

    def step():
        model.apply_sgd(learning_rate, mu)
    return step


def nesterov_momentum(model, learning_rate, momentum=0.9):
    mu = momentum

    def step():
        model.apply_sgd(learning_rate, mu, nesterov=True)
    return step


def adadelta(model, learning_rate=1.0, rho=0.95, epsilon=1e-6):
    def step():
        model.apply_adadelta(learning_rate, rho, epsilon)
    return step
