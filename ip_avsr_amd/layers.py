"""The handful of ``lasagne.layers`` helper functions the reference scripts call on a network
(runners/3stream.py:304-305,393,425; cuave/bimodal_with_val.py:349), forwarding to the model object."""


class FuseHandle(object):
    """Stands in for the merge layer ``l_fuse`` returned next to the network by the N-stream factories."""

    def __init__(self, model, name):
        self.model, self.name = model, name

    def get_all_param_values(self, **tags):
        return self.model.get_all_param_values(**tags)


def _model(obj):
    return getattr(obj, "model", obj)


def get_all_params(network, **tags):
    return _model(network).get_all_params(**tags)


def get_all_param_values(network, **tags):
    return _model(network).get_all_param_values(**tags)


def set_all_param_values(network, values, **tags):
    return _model(network).set_all_param_values(values, **tags)


def count_params(network, **tags):
    return _model(network).count_params()


def get_all_layer_names(network):
    """Names of the parameterised layers in topological order (what the extractors walk,
    modelzoo/deltanet_majority_vote.py:145-196)."""
    seen, out = set(), []
    for p in _model(network).params:
        layer = p.name.rsplit(".", 1)[0]
        if layer not in seen:
            seen.add(layer)
            out.append(layer)
    return out
