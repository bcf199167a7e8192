"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's local contrast normalisation (utils/lcn.py).

Only tests/ (and the CPU-baseline legs of the profile scripts) may import this module; the product path
(ip_avsr_amd/utils/lcn.py -> csrc/prep.hip) never does.

PARITY UNPINNED: the reference builds this op out of Theano's conv2d, which is not installable here, and holds no
test or fixture for it.  What pins this restatement instead (tests/test_lcn_oracle.py): the 'full' convolution cropped
by mid = k // 2 is checked against scipy.signal.convolve2d and against torch's conv2d with a flipped filter and zero
padding; closed-form cases (constant image interior, threshold-dominated divisor); the column-mean quirk of
`denom.mean(axis=[1, 2])` on a (B, 1, H, W) tensor is restated literally with numpy on that 4-d shape.
"""
import numpy as np


def gaussian_filter(kernel_shape):
    """utils/lcn.py:9-21: float32 array filled element by element, then divided by its sum."""
    k = int(kernel_shape)
    x = np.zeros((k, k), dtype="float32")
    mid = np.floor(k / 2.)
    sigma = 2.0
    for i in range(k):
        for j in range(k):
            Z = 2 * np.pi * sigma ** 2
            x[i, j] = 1. / Z * np.exp(-((i - mid) ** 2 + (j - mid) ** 2) / (2. * sigma ** 2))
    return x / np.sum(x)


def conv_full_cropped(X, f):
    """theano conv2d(border_mode='full') of a (B, 1, H, W) batch with one (1, 1, k, k) filter -- a true convolution,
    out[y][x] = sum_ij f[i][j] X[y - i][x - j] -- followed by the crop [mid:-mid] in both image axes
    (utils/lcn.py:72-80).  Returns (B, 1, H, W)."""
    B, _, H, W = X.shape
    k = f.shape[0]
    mid = int(np.floor(k / 2.))
    full = np.zeros((B, 1, H + k - 1, W + k - 1), dtype=X.dtype)
    for i in range(k):
        for j in range(k):
            full[:, :, i:i + H, j:j + W] += f[i, j] * X
    return full[:, :, mid:-mid, mid:-mid] if mid else full


def lecun_lcn(X, img_shape, kernel_shape, threshold=1e-4, dtype=np.float64, filt=None):
    """utils/lcn.py:64-104 (make_lecun_lcn) applied to X, any array of B*H*W values; returns (B, H, W).
    ``filt``: another k x k filter in place of the Gaussian (tests of the C ABI's generic path)."""
    H, W = int(img_shape[0]), int(img_shape[1])
    X = np.asarray(X, dtype=dtype).reshape(-1, 1, H, W)
    f = (gaussian_filter(kernel_shape) if filt is None else np.asarray(filt)).astype(dtype)
    centered = X - conv_full_cropped(X, f)                                   # :76-80
    denom = np.sqrt(conv_full_cropped(centered ** 2, f))                     # :83-89
    per_img_mean = denom.mean(axis=(1, 2))                                   # :90  -> (B, W): mean over channel AND rows
    divisor = np.maximum(per_img_mean[:, None, None, :], denom)              # :91  dimshuffle(0, 'x', 'x', 1)
    divisor = np.maximum(divisor, threshold)                                 # :92
    new_X = centered / divisor                                               # :94
    return new_X.transpose(0, 2, 3, 1).reshape(X.shape[0], H, W)             # :95-96 (one channel)
