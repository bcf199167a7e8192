"""The conv auto-encoder with DropoutLayers (0.2 on the input, 0.5 behind the poolings, the flatten and the dense layer)
and the widths divided by the keep probabilities: 125 / 300 / 400 filters, 2 x DENSE, 2 x BOTTLENECK
(modelzoo/avletters_convae_drop.py:33-75)."""
from ..convae import ConvAE
from .avletters_convae import _Encoder


def create_model(incoming, options):
    """``incoming``: the input shape ``(None, 1, H, W)`` (or a layer-like object with ``.shape`` / ``.output_shape``);
    ``options``: {'BOTTLENECK': 50, 'DENSE': 500[, 'PRECISION': 'f32' | 'bf16']}."""
    shape = getattr(incoming, "output_shape", getattr(incoming, "shape", incoming))
    ae = ConvAE((int(shape[-2]), int(shape[-1])), int(options['DENSE'] / 0.5), int(options['BOTTLENECK'] / 0.5), options.get('PRECISION', 'f32'), variant='dropout')
    ae.init_params()
    return ae, _Encoder(ae)
