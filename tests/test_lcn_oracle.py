"""Pins oracle/lcn_oracle.py (the restatement of reference utils/lcn.py; the Theano original cannot run here):
its convolution against scipy and torch, closed-form cases, and the host-side filter of the product module."""
import numpy as np
import pytest

from oracle import lcn_oracle as L


def test_gaussian_filter_is_normalised_and_matches_the_product_module():
    from ip_avsr_amd.utils import lcn as P
    for k in (3, 7, 9):
        f = L.gaussian_filter(k)
        assert f.dtype == np.float32 and f.shape == (k, k)
        assert abs(float(f.sum()) - 1.0) < 1e-6
        assert f[k // 2, k // 2] == f.max() and np.allclose(f, f.T) and np.allclose(f, f[::-1, ::-1])
        np.testing.assert_array_equal(P.gaussian_filter(k), f)
    with pytest.raises(ValueError):
        P.make_lecun_lcn((2, 1, 8, 8), (8, 8), 4)


def test_cropped_full_convolution_matches_scipy_and_torch():
    from scipy.signal import convolve2d
    import torch
    rng = np.random.default_rng(3)
    for (H, W, k) in ((9, 7, 3), (30, 40, 9), (12, 5, 7)):
        X = rng.normal(size=(3, 1, H, W))
        f = rng.normal(size=(k, k))                       # asymmetric on purpose: convolution, not correlation
        got = L.conv_full_cropped(X, f)
        mid = k // 2
        for b in range(3):
            ref = convolve2d(X[b, 0], f, mode="full")[mid:-mid, mid:-mid]
            np.testing.assert_allclose(got[b, 0], ref, rtol=1e-12, atol=1e-12)
        t = torch.nn.functional.conv2d(torch.tensor(X), torch.tensor(f[::-1, ::-1].copy())[None, None], padding=mid)
        np.testing.assert_allclose(got, t.numpy(), rtol=1e-10, atol=1e-10)


def test_closed_forms():
    # constant image: away from the border blur(X) = X (up to the float32 filter's sum), so the centred image is ~0
    # there; at the border it is not
    X = np.full((1, 20, 24), 3.0)
    out = L.lecun_lcn(X, (20, 24), 7, threshold=1e-4)
    assert out.shape == (1, 20, 24)
    assert np.abs(out[0, 3:-3, 3:-3]).max() < 1e-5
    assert np.abs(out[0, 0]).min() > 0
    # a threshold above every local norm divides the centred image by the threshold
    rng = np.random.default_rng(0)
    X = rng.normal(size=(2, 10, 12))
    big = L.lecun_lcn(X, (10, 12), 5, threshold=100.0)
    f = L.gaussian_filter(5).astype(np.float64)
    cen = X.reshape(2, 1, 10, 12) - L.conv_full_cropped(X.reshape(2, 1, 10, 12), f)
    np.testing.assert_allclose(big, cen[:, 0] / 100.0, rtol=1e-12)
    # the divisor floor is the COLUMN mean of the local norm (mean over rows), not an image mean
    X = rng.normal(size=(1, 8, 6)); X[0, :, 0] *= 50.0
    den = np.sqrt(L.conv_full_cropped((X.reshape(1, 1, 8, 6) - L.conv_full_cropped(X.reshape(1, 1, 8, 6), L.gaussian_filter(3).astype(float))) ** 2,
                                      L.gaussian_filter(3).astype(float)))[0, 0]
    out = L.lecun_lcn(X, (8, 6), 3, threshold=1e-9)
    cen = (X.reshape(1, 1, 8, 6) - L.conv_full_cropped(X.reshape(1, 1, 8, 6), L.gaussian_filter(3).astype(float)))[0, 0]
    np.testing.assert_allclose(out[0], cen / np.maximum(den.mean(axis=0)[None, :], den), rtol=1e-12)
    # scale invariance above the threshold: lcn(a X) = lcn(X)
    np.testing.assert_allclose(L.lecun_lcn(7.0 * X, (8, 6), 3, threshold=1e-9), out, rtol=1e-10)
