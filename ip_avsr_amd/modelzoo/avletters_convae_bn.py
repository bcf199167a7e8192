"""The conv auto-encoder with BatchNormLayers behind both poolings, on the flattened conv output and behind the dense
layer (modelzoo/avletters_convae_bn.py:33-74; layer names conv2d1, batchnorm2, conv2d4, batchnorm3, conv2d7, batchnorm8,
dense10, batchnorm11, bottleneck, dense12, dense13, deconv2d19, deconv2d17, deconv2d14)."""
from ..convae import ConvAE
from .avletters_convae import _Encoder


def create_model(incoming, options):
    """``incoming``: the input shape ``(None, 1, H, W)`` (or a layer-like object with ``.shape`` / ``.output_shape``);
    ``options``: {'BOTTLENECK': 50, 'DENSE': 500[, 'PRECISION': 'f32' | 'bf16']}."""
    shape = getattr(incoming, "output_shape", getattr(incoming, "shape", incoming))
    ae = ConvAE((int(shape[-2]), int(shape[-1])), options['DENSE'], options['BOTTLENECK'], options.get('PRECISION', 'f32'), variant='batchnorm')
    ae.init_params()
    return ae, _Encoder(ae)
