"""LeCun local contrast normalisation (reference utils/lcn.py) on the GPU.

Same three names as the reference module.  ``make_lecun_lcn`` / ``lecun_lcn`` return a callable, as the reference
returns a compiled Theano function; the callable runs ``csrc/prep.hip::lcn_kernel`` through the C ABI
(``adn_prep_lcn``).  NumPy in -> NumPy out, CUDA torch tensor in -> CUDA tensor out; fails loudly without the HIP
library or a GPU.
"""
import ctypes as C

import numpy as np

from .. import _lib


def gaussian_filter(kernel_shape):
    """Normalised kernel_shape x kernel_shape Gaussian, sigma = 2, float32 (utils/lcn.py:9-21)."""
    k = int(kernel_shape)
    mid = np.floor(k / 2.)
    d = np.arange(k) - mid
    yy, xx = np.meshgrid(d, d, indexing="ij")
    sigma = 2.0
    x = (1. / (2 * np.pi * sigma ** 2) * np.exp(-(yy ** 2 + xx ** 2) / (2. * sigma ** 2))).astype(np.float32)
    return x / np.sum(x)


def _make(img_shape, kernel_shape, threshold):
    H, W = int(img_shape[0]), int(img_shape[1])
    k = int(kernel_shape)
    if k % 2 == 0:                      # the reference's crop [mid:-mid] of the 'full' convolution only fits odd sizes
        raise ValueError("kernel_shape must be odd")
    filt = np.ascontiguousarray(gaussian_filter(k), dtype=np.float32)
    lib = _lib.load()

    def f(X):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("ip_avsr_amd.utils.lcn needs a GPU")
        was_numpy = not isinstance(X, torch.Tensor)
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(X, dtype=np.float32)), device="cuda") if was_numpy \
            else X.to(torch.float32).contiguous()
        if not t.is_cuda:
            raise ValueError("torch inputs must be CUDA tensors")
        if t.numel() % (H * W):
            raise ValueError("input does not hold whole %dx%d images" % (H, W))
        n = t.numel() // (H * W)
        out = torch.empty((n, H, W), dtype=torch.float32, device=t.device)
        _lib.check(lib.adn_prep_lcn(C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), n, H, W,
                                    filt.ctypes.data_as(C.c_void_p), k, C.c_float(float(threshold)),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out.cpu().numpy() if was_numpy else out

    return f


def lecun_lcn(input, img_shape, kernel_shape, threshold=1e-4):
    """utils/lcn.py:24-61: returns the function (of a (batch, rows*cols) matrix); ``input`` only fixes the batch."""
    return _make(img_shape, kernel_shape, threshold)


def make_lecun_lcn(input_shape, img_shape, kernel_shape, threshold=1e-4):
    """utils/lcn.py:64-104: ``input_shape`` = (batch, 1, rows, cols); returns f(X) -> (batch, rows, cols)."""
    if len(input_shape) == 4 and input_shape[1] not in (1, None):
        raise ValueError("local contrast normalisation is defined for one channel (the filter shape is (1, 1, k, k))")
    return _make(img_shape, kernel_shape, threshold)
