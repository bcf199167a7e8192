"""Calibration only (not a product path): what the vendor library's plain bf16 GEMM reaches on this box for the
encoder shapes of the bench geometry, to put the hand-written kernels' TFLOP/s in context.
Usage: python profiles/hipblaslt_calibration.py > profiles/r02/hipblaslt_calibration.txt"""
import torch

def bench(m, n, k, trans_a=False, iters=30):
    dev = "cuda"
    a = torch.randn((k, m) if trans_a else (m, k), device=dev, dtype=torch.bfloat16)
    b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
    f = (lambda: a.t() @ b) if trans_a else (lambda: a @ b)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * m * n * k / us / 1e6

if __name__ == "__main__":
    print("%-28s %3s %6s %6s %6s | %9s %9s" % ("case", "lay", "M", "N", "K", "us", "TFLOP/s"))
    R = 20800
    for name, ta, m, n, k in [("fwd fc1", 0, R, 2000, 1200), ("fwd fc2", 0, R, 1000, 2000), ("fwd fc3", 0, R, 500, 1000),
                              ("dX fc2", 0, R, 2000, 1000), ("x3 fwd fc1", 0, 3 * R, 2000, 1200), ("x3 fwd fc2", 0, 3 * R, 1000, 2000),
                              ("dW fc1", 1, 1200, 2000, R), ("dW fc2", 1, 2000, 1000, R), ("dW fc3", 1, 1000, 500, R),
                              ("square 8192", 0, 8192, 8192, 8192)]:
        us, tf = bench(m, n, k, bool(ta))
        print("%-28s %3s %6d %6d %6d | %9.1f %9.1f" % (name, "TN" if ta else "NN", m, n, k, us, tf))
