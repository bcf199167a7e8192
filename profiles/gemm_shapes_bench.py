"""Per-shape GEMM micro-benchmark: every (layout, M, N, K) the AVLetters trimodal train step launches at
B=520, T=40, timed with torch events over repeated launches, for the three arithmetic paths
(f32 MFMA, bf16 converting in flight, bf16 with pre-converted shadows).  Prints TFLOP/s per shape.

    python profiles/gemm_shapes_bench.py            (on an MI355X)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ip_avsr_amd import _lib

lib = _lib.load()
R = 20800
SHAPES = []
dims = [1200, 2000, 1000, 500, 50]
for a, b in zip(dims[:-1], dims[1:]):
    SHAPES.append(("fwd  NN", 0, R, b, a))
for a, b in zip(dims[1:-1], dims[2:]):
    SHAPES.append(("dX   NT", 1, R, a, b))
for a, b in zip(dims[:-1], dims[1:]):
    SHAPES.append(("dW   TN", 2, a, b, R))
SHAPES += [("xproj NN", 0, R, 1000, 150), ("aggx  NN", 0, R, 1000, 250), ("dWin  TN", 2, 150, 1000, R),
           ("dWhid TN", 2, 250, 1000, R), ("dfeat NT", 1, R, 150, 1000), ("dh    NT", 1, R, 250, 1000),
           ("cls   NN", 0, R, 26, 250)]
pad = lambda n: (n + 7) // 8 * 8


def dptr(t):
    return C.c_void_p(t.data_ptr())


def time_it(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


ONLY = os.environ.get("GEMM_ONLY")          # e.g. "fwd  NN:1" -> only the 2nd shape named so, shadow variant only
if ONLY:
    nm, idx = ONLY.split(":")
    SHAPES = [[s for s in SHAPES if s[0] == nm][int(idx)]]
print("%-10s %6s %6s %6s | %9s %9s %9s   (TFLOP/s)" % ("shape", "M", "N", "K", "f32", "bf16-cvt", "bf16-shdw"))
tot = [0.0, 0.0, 0.0]
for name, layout, M, N, K in SHAPES:
    ash = (M, pad(K)) if layout != 2 else (K, pad(M))
    bsh = (K, pad(N)) if layout != 1 else (N, pad(K))
    A = torch.randn(*ash, device="cuda"); Bm = torch.randn(*bsh, device="cuda")
    Cm = torch.zeros(M, pad(N), device="cuda")
    A16 = torch.empty(ash, device="cuda", dtype=torch.bfloat16); B16 = torch.empty(bsh, device="cuda", dtype=torch.bfloat16)
    C16 = torch.empty(M, pad(N), device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.adn_op_to_bf16(dptr(A), dptr(A16), A.numel(), None))
    _lib.check(lib.adn_op_to_bf16(dptr(Bm), dptr(B16), Bm.numel(), None))
    acc = 1 if layout == 2 else 0
    f = [lambda: lib.adn_op_gemm_ex(layout, M, N, K, dptr(A), ash[1], dptr(Bm), bsh[1], dptr(Cm), pad(N), None, 0, acc, 0, None),
         lambda: lib.adn_op_gemm_ex(layout, M, N, K, dptr(A), ash[1], dptr(Bm), bsh[1], dptr(Cm), pad(N), None, 0, acc, 1, None),
         lambda: lib.adn_op_gemm_shadow(layout, M, N, K, dptr(A), ash[1], dptr(Bm), bsh[1], dptr(Cm), pad(N), dptr(A16),
                                        dptr(B16), dptr(C16) if layout != 2 else None, acc, None)]
    if ONLY:
        f = [f[2], f[2], f[2]]
    ms = [time_it(fn) for fn in f]
    fl = 2.0 * M * N * K
    for i in range(3):
        tot[i] += ms[i]
    print("%-10s %6d %6d %6d | %9.1f %9.1f %9.1f   ms %.3f %.3f %.3f" % ((name, M, N, K) + tuple(fl / (m * 1e-3) / 1e12 for m in ms) + tuple(ms)))
print("sum of one launch each (ms):", tot)
