"""Delta / acceleration coefficients of a feature sequence (reference utils/signal.py:42-80) on the GPU.

The reference builds these out of ``theano.scan``; here they are one launch of the delta-layer kernel
(``csrc/elementwise.hip``, ``adn_op_delta_forward``) -- the same kernel the models' DeltaLayer uses.  Same names and
argument meaning: ``A`` is ``(time_steps, features)`` (a batch ``(B, T, F)`` is accepted too); the sequence is padded
by repeating its first / last frame ``theta`` times and ``delta[t] = sum_k (A[t+k] - A[t-k]) / (2 k)``.
NumPy in -> NumPy out, CUDA tensor in -> CUDA tensor out; fails loudly without the HIP library or a GPU."""
import ctypes as C

import numpy as np

from .. import _lib


def _run(A, theta):
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("ip_avsr_amd.utils.signal needs a GPU")
    was_numpy = not isinstance(A, torch.Tensor)
    t = torch.as_tensor(np.ascontiguousarray(np.asarray(A, dtype=np.float32)), device="cuda") if was_numpy \
        else A.to(torch.float32).contiguous()
    if not t.is_cuda:
        raise ValueError("torch inputs must be CUDA tensors")
    single = t.dim() == 2
    if single:
        t = t[None]
    if t.dim() != 3:
        raise ValueError("expected (time_steps, features) or (batch, time_steps, features)")
    B, T, F = (int(d) for d in t.shape)
    out = torch.empty((T, B, 3 * F), dtype=torch.float32, device=t.device)
    _lib.check(_lib.load().adn_op_delta_forward(C.c_void_p(t.data_ptr()), F, C.c_void_p(out.data_ptr()), 3 * F, B, T, F,
                                                int(theta), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    out = out.permute(1, 0, 2)                      # (B, T, [A | delta | delta-delta])
    if single:
        out = out[0]
    return out, F, was_numpy


def delta_coeff(A, theta):
    """utils/signal.py:42-56: the delta coefficients, same shape as ``A``."""
    out, F, was_numpy = _run(A, theta)
    d = out[..., F:2 * F].contiguous()
    return d.cpu().numpy() if was_numpy else d


def append_delta_coeff(A, theta):
    """utils/signal.py:59-80: ``[A | delta | delta-delta]`` along the feature axis (delta-delta = the same operator applied
    to the re-padded deltas)."""
    out, F, was_numpy = _run(A, theta)
    out = out.contiguous()
    return out.cpu().numpy() if was_numpy else out
