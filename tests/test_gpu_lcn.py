"""Local contrast normalisation on the GPU (csrc/prep.hip::lcn_kernel through adn_prep_lcn and the reference-named
wrappers of ip_avsr_amd/utils/lcn.py) against the NumPy restatement of reference utils/lcn.py (oracle/lcn_oracle.py).
fp32 on the GPU, fp64 in the oracle: 2e-5 of the output scale."""
import numpy as np
import pytest

from oracle import lcn_oracle as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ip_avsr_amd.utils import lcn
    return lcn


@pytest.mark.parametrize("H,W,k,thr,B", [(30, 40, 9, 1e-4, 37), (26, 44, 7, 10.0, 5), (30, 50, 9, 1e-2, 3), (5, 3, 3, 1e-4, 2),
                                         (64, 128, 15, 1e-3, 2), (1, 1, 1, 1e-4, 4)])
def test_lcn_matches_oracle(P, H, W, k, thr, B):
    rng = np.random.default_rng(H * W + k)
    X = (rng.normal(size=(B, 1, H, W)) * rng.uniform(0.5, 20.0, size=(B, 1, 1, 1))).astype(np.float32)
    ref = L.lecun_lcn(X, (H, W), k, thr)
    out = P.make_lecun_lcn((B, 1, H, W), (H, W), k, thr)(X)
    assert out.shape == (B, H, W) and out.dtype == np.float32
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    out2 = P.lecun_lcn(X.reshape(B, -1), (H, W), k, thr)(X.reshape(B, -1))          # the matrix-input spelling
    np.testing.assert_array_equal(out2, out)


def test_lcn_device_tensor_in_place_of_numpy_and_errors(P):
    import torch
    rng = np.random.default_rng(1)
    X = rng.normal(size=(6, 1, 30, 40)).astype(np.float32)
    f = P.make_lecun_lcn((6, 1, 30, 40), (30, 40), 9)
    a = f(X)
    b = f(torch.tensor(X, device="cuda"))
    assert isinstance(b, torch.Tensor) and b.is_cuda
    np.testing.assert_array_equal(b.cpu().numpy(), a)
    with pytest.raises(ValueError):
        f(X[:, :, :-1])                                   # not whole images
    with pytest.raises(RuntimeError):
        P.make_lecun_lcn((1, 1, 100, 100), (100, 100), 9)(np.zeros((1, 1, 100, 100), np.float32))   # > 8192 pixels
    # scale invariance above the threshold: the divisor scales with the image
    c = f(7.0 * X)
    assert np.abs(c - a).max() < 1e-4 * np.abs(a).max()


def test_c_abi_with_a_filter_that_is_not_separable(P):
    """adn_prep_lcn takes any odd k x k filter; only rank-1 filters take the two-pass route."""
    import ctypes as C
    import torch
    from ip_avsr_amd import _lib
    rng = np.random.default_rng(5)
    B, H, W, k = 4, 17, 23, 5
    X = rng.normal(size=(B, 1, H, W)).astype(np.float32)
    filt = np.abs(rng.normal(size=(k, k))).astype(np.float32)
    filt /= filt.sum()
    x = torch.tensor(X, device="cuda")
    y = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    lib = _lib.load()
    _lib.check(lib.adn_prep_lcn(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, H, W, filt.ctypes.data_as(C.c_void_p), k,
                                C.c_float(1e-3), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ref = L.lecun_lcn(X, (H, W), k, 1e-3, filt=filt)
    assert np.abs(y.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    with pytest.raises(Exception):
        _lib.check(lib.adn_prep_lcn(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, H, W, filt.ctypes.data_as(C.c_void_p), 4,
                                    C.c_float(1e-3), None))
