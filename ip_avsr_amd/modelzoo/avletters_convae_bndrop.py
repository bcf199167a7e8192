"""The conv auto-encoder with dropout and BatchNormLayers behind every convolution's nonlinearity and the dense layer,
ScaledTanh(2/3, 1.7159) (modelzoo/avletters_convae_bndrop.py:8,33-77)."""
from ..convae import ConvAE
from .avletters_convae import _Encoder


def create_model(incoming, options):
    """``incoming``: the input shape ``(None, 1, H, W)`` (or a layer-like object with ``.shape`` / ``.output_shape``);
    ``options``: {'BOTTLENECK': 50, 'DENSE': 500[, 'PRECISION': 'f32' | 'bf16']}."""
    shape = getattr(incoming, "output_shape", getattr(incoming, "shape", incoming))
    ae = ConvAE((int(shape[-2]), int(shape[-1])), options['DENSE'], options['BOTTLENECK'], options.get('PRECISION', 'f32'), variant='bn+dropout')
    ae.init_params()
    return ae, _Encoder(ae)
