"""The feature front-end of ``utils/preprocessing.py`` on the GPU.

Same names and argument meaning as the reference's NumPy functions; the arithmetic runs in the HIP kernels of
``csrc/prep.hip`` through the C ABI (``adn_prep_*``), in fp32 (the reference keeps float64 where its input is
float64; see tests/test_gpu_prep.py for the tolerances).  Inputs may be NumPy arrays (copied to the device, result
returned as NumPy) or CUDA torch tensors (used in place, result returned as a CUDA tensor on the same device) -- torch
is only the device allocator here.  Fails loudly when the HIP library or a GPU is missing; the NumPy versions live in
``utils/preprocessing.py``.
"""
import ctypes as C

import numpy as np

from .. import _lib
from . import preprocessing as _host


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("ip_avsr_amd.utils.preprocessing_gpu needs a GPU (use utils.preprocessing on the host)")
    return torch


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Frames(object):
    """Device copy of a (frames, D) matrix plus how to hand the result back."""

    def __init__(self, x):
        torch = _torch()
        self.torch = torch
        self.was_numpy = not isinstance(x, torch.Tensor)
        if self.was_numpy:
            x = torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32)), device="cuda")
        else:
            if not x.is_cuda:
                raise ValueError("torch inputs must be CUDA tensors")
            x = x.to(torch.float32).contiguous()
        if x.dim() != 2:
            raise ValueError("expected a (frames, features) matrix")
        self.t = x

    def back(self, t):
        return t.cpu().numpy() if self.was_numpy else t


_INDEX_CACHE = {}


def _utt_index(torch, lens, n_frames, device):
    """(first, last, starts, lens) int32 device vectors for a length vector; cached (a dataset's transforms all share
    one length vector, and building + uploading it costs more than the kernels)."""
    lens = np.asarray(lens).reshape(-1).astype(np.int64)
    key = (lens.tobytes(), str(device))
    hit = _INDEX_CACHE.get(key)
    if hit is not None and n_frames == hit[4]:
        return hit[:4]
    out = _utt_index_build(torch, lens, n_frames, device)
    if len(_INDEX_CACHE) > 8:
        _INDEX_CACHE.clear()
    _INDEX_CACHE[key] = out + (n_frames,)
    return out


def _utt_index_build(torch, lens, n_frames, device):
    if lens.sum() != n_frames:
        raise ValueError("sequence lengths sum to %d, matrix has %d frames" % (lens.sum(), n_frames))
    if (lens < 1).any():
        raise ValueError("empty utterance")
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    first = np.repeat(starts, lens).astype(np.int32)
    last = np.repeat(starts + lens - 1, lens).astype(np.int32)
    dev = lambda a: torch.as_tensor(a, device=device)
    return dev(first), dev(last), dev(starts.astype(np.int32)), dev(lens.astype(np.int32))


def concat_first_second_deltas(X, vidlenvec, w=9):
    """[X | deltas | deltas-of-deltas] per utterance (reference utils/preprocessing.py:465-489)."""
    lib = _lib.load()
    f = _Frames(X)
    torch, x = f.torch, f.t
    n, F = x.shape
    first, last, _, _ = _utt_index(torch, vidlenvec, n, x.device)
    y = torch.empty(n, 3 * F, device=x.device, dtype=torch.float32)
    y[:, :F] = x
    s = _stream(torch)
    d1 = y[:, F:2 * F]
    d2 = y[:, 2 * F:]
    _lib.check(lib.adn_prep_seq_deltas(_ptr(x), F, _ptr(d1), 3 * F, _ptr(first), _ptr(last), n, F, int(w), s))
    _lib.check(lib.adn_prep_seq_deltas(_ptr(d1), 3 * F, _ptr(d2), 3 * F, _ptr(first), _ptr(last), n, F, int(w), s))
    return f.back(y)


def deltas(x, w=9):
    """Deltas of each ROW of ``x`` along its columns (reference utils/preprocessing.py:17-51): one "utterance" whose
    frames are the columns."""
    f = _Frames(np.asarray(x).T if not hasattr(x, "is_cuda") else x.t())
    torch, xt = f.torch, f.t
    lib = _lib.load()
    n, F = xt.shape
    first, last, _, _ = _utt_index(torch, [n], n, xt.device)
    out = torch.empty_like(xt)
    _lib.check(lib.adn_prep_seq_deltas(_ptr(xt), F, _ptr(out), F, _ptr(first), _ptr(last), n, F, int(w), _stream(torch)))
    return f.back(out.t().contiguous())


def compute_diff_images(X, vidlenvec):
    """Frame differences per utterance (reference utils/preprocessing.py:506-517)."""
    lib = _lib.load()
    f = _Frames(X)
    torch, x = f.torch, f.t
    n, D = x.shape
    first, last, _, _ = _utt_index(torch, vidlenvec, n, x.device)
    out = torch.empty_like(x)
    _lib.check(lib.adn_prep_diff_images(_ptr(x), _ptr(out), D, _ptr(first), _ptr(last), n, D, _stream(torch)))
    return f.back(out)


def sequencewise_mean_image_subtraction(input, seqlens, axis=0):
    """Subtract each utterance's mean frame (reference utils/preprocessing.py:260-277)."""
    if axis != 0:
        raise NotImplementedError("only axis=0 (mean over the frames of an utterance), as every caller uses it")
    lib = _lib.load()
    f = _Frames(input)
    torch, x = f.torch, f.t
    n, D = x.shape
    _, _, starts, lens = _utt_index(torch, seqlens, n, x.device)
    out = torch.empty_like(x)
    _lib.check(lib.adn_prep_mean_image_subtraction(_ptr(x), _ptr(out), D, _ptr(starts), _ptr(lens), int(lens.numel()), D,
                                                   _stream(torch)))
    return f.back(out)


def normalize_input(input, centralize=True, quantize=False):
    """Per-frame z-normalisation, population std (reference utils/preprocessing.py:218-242).  A CUDA tensor is
    normalised in place like the reference's array; a NumPy input is returned as a new array."""
    if quantize or not centralize:
        raise NotImplementedError("GPU path covers centralize=True, quantize=False (what every runner passes)")
    lib = _lib.load()
    f = _Frames(input)
    torch, x = f.torch, f.t
    n, D = x.shape
    _lib.check(lib.adn_prep_normalize_rows(_ptr(x), D, n, D, _stream(torch)))
    if not f.was_numpy and x.data_ptr() != input.data_ptr():
        input.copy_(x)
        return input
    return f.back(x)


def featurewise_normalize_sequence(input):
    """Column z-normalisation; returns (normalised, mean, std) (reference utils/preprocessing.py:245-257)."""
    lib = _lib.load()
    f = _Frames(input)
    torch, x = f.torch, f.t
    n, D = x.shape
    ws = torch.empty(2 * D, device=x.device, dtype=torch.float64)
    mean = torch.empty(D, device=x.device, dtype=torch.float32)
    std = torch.empty(D, device=x.device, dtype=torch.float32)
    s = _stream(torch)
    _lib.check(lib.adn_prep_column_stats(_ptr(x), D, n, D, _ptr(ws), _ptr(mean), _ptr(std), s))
    out = torch.empty_like(x)
    _lib.check(lib.adn_prep_apply_column_norm(_ptr(x), _ptr(out), D, n, D, _ptr(mean), _ptr(std), s))
    return f.back(out), f.back(mean), f.back(std)


def apply_featurewise_normalization(input, mean, std):
    """(input - mean) / std with the statistics of another split (reference runners/3stream.py:102-108)."""
    lib = _lib.load()
    f = _Frames(input)
    torch, x = f.torch, f.t
    n, D = x.shape
    dev = lambda a: a.to(x.device, torch.float32).contiguous() if isinstance(a, torch.Tensor) else \
        torch.as_tensor(np.asarray(a, np.float32), device=x.device)
    mean, std = dev(mean), dev(std)
    out = torch.empty_like(x)
    _lib.check(lib.adn_prep_apply_column_norm(_ptr(x), _ptr(out), D, n, D, _ptr(mean), _ptr(std), _stream(torch)))
    return f.back(out)


def reorder_data(X, shape, orig_order="f", desired_order="c"):
    """Re-pack flattened (d1,d2) images between Fortran and C pixel order (reference utils/preprocessing.py:492-503):
    a column permutation."""
    d1, d2 = shape
    f = _Frames(np.asarray(X).reshape((-1, d1 * d2)) if not hasattr(X, "is_cuda") else X.reshape(-1, d1 * d2))
    torch, x = f.torch, f.t
    if orig_order.lower() == desired_order.lower():
        return f.back(x)
    lib = _lib.load()
    # the permutation the host function applies, read off an index image (one frame is enough: it acts per frame)
    perm = _host.reorder_data(np.arange(d1 * d2, dtype=np.int64)[None, :], shape, orig_order, desired_order)[0]
    perm_d = torch.as_tensor(perm.astype(np.int32), device=x.device)
    n, D = x.shape
    out = torch.empty_like(x)
    _lib.check(lib.adn_prep_gather_columns(_ptr(x), D, _ptr(out), D, _ptr(perm_d), n, D, _stream(torch)))
    return f.back(out)


def dct_basis(D, columns):
    """Orthonormal DCT-II basis vectors (scipy.fftpack.dct(norm='ortho')) for the requested coefficient indices, as a
    (D, len(columns)) float32 matrix: coefficient k of x is x @ basis[:, k]."""
    n = np.arange(D, dtype=np.float64)[:, None]
    k = np.asarray(columns, dtype=np.float64)[None, :]
    scale = np.where(k == 0, np.sqrt(1.0 / D), np.sqrt(2.0 / D))
    return (scale * np.cos(np.pi * (2.0 * n + 1.0) * k / (2.0 * D))).astype(np.float32)


def compute_dct_features(X, image_shape, no_coeff=30, method="zigzag"):
    """DCT features (reference utils/preprocessing.py:417-462): a 1-D orthonormal DCT-II over the flattened image
    (quirk kept), zig-zag selection skipping DC.  Only the selected coefficients are computed: one
    (frames x D) x (D x no_coeff) product on the fp32 MFMA GEMM of the training path."""
    if method != "zigzag":
        raise NotImplementedError("GPU path covers method='zigzag' (utils.preprocessing has the score-based selections)")
    lib = _lib.load()
    f = _Frames(X)
    torch, x = f.torch, f.t
    n, D = x.shape
    r, c = zip(*_host._zigzag_order(*image_shape)[1:no_coeff + 1])
    flat_idx = np.array(r) * image_shape[1] + np.array(c)
    K = len(flat_idx)
    pad = lambda v: (v + 3) // 4 * 4
    basis = torch.zeros(D, pad(K), device=x.device, dtype=torch.float32)
    basis[:, :K] = torch.as_tensor(dct_basis(D, flat_idx), device=x.device)
    if D % 4:
        xa = torch.zeros(n, pad(D), device=x.device, dtype=torch.float32)
        xa[:, :D] = x
    else:
        xa = x
    out = torch.empty(n, pad(K), device=x.device, dtype=torch.float32)
    _lib.check(lib.adn_op_gemm(0, n, K, D, _ptr(xa), xa.shape[1], _ptr(basis), pad(K), _ptr(out), pad(K), None, 0, 0,
                               _stream(torch)))
    return f.back(out[:, :K].contiguous())
