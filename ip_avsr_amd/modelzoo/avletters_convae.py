"""The convolutional auto-encoder of the reference (modelzoo/avletters_convae.py:33-69): ``create_model(incoming,
options)`` returns ``(network, bottleneck)`` -- here two views of one ``ip_avsr_amd.convae.ConvAE``: the network
reconstructs (``recon_fn``, ``train``, ``cost``), the bottleneck view encodes."""
from ..convae import ConvAE


class _Encoder(object):
    """The 'bottleneck' layer handle: get_output == the 50-d code."""

    def __init__(self, ae):
        self.ae = ae

    def __call__(self, x):
        return self.ae.encode(x)

    def get_all_param_values(self):
        # get_all_param_values(bottleneck): every layer up to the bottleneck, BatchNorm parameters included
        names = self.ae.param_names[:self.ae.param_names.index("bottleneck.b") + 1]
        return [self.ae.get_param(n) for n in names]


def create_model(incoming, options):
    """``incoming``: the input shape ``(None, 1, H, W)`` (or a layer-like object with ``.shape`` / ``.output_shape``);
    ``options``: {'BOTTLENECK': 50, 'DENSE': 500[, 'PRECISION': 'f32' | 'bf16']}."""
    shape = getattr(incoming, "output_shape", getattr(incoming, "shape", incoming))
    ae = ConvAE((int(shape[-2]), int(shape[-1])), options['DENSE'], options['BOTTLENECK'], options.get('PRECISION', 'f32'))
    ae.init_params()
    return ae, _Encoder(ae)
