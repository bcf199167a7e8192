"""Conv auto-encoder training-step throughput and MFMA utilisation, beside the NumPy oracle on the host.

One step = forward + mean-squared error + backward + adadelta on a batch of 128 frames of 30x40 (the reference's batch,
avletters/avletters_convae.py:276) resident in HBM.  FLOPs counted = 2 M N K of every GEMM of the layer-by-layer formulation (im2col /
col2im data movement is not arithmetic; since round 2 the two Upscale2DLayers are folded into the deconvolutions behind them,
which runs those layers' three GEMMs on a quarter of the rows -- the count stays the reference formulation's, so the TFLOP/s
figure is an effective one).

    python profiles/convae_bench.py        (on an MI355X)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ip_avsr_amd.convae import ConvAE
from oracle import convae_oracle as CO

HW = (30, 40)
g = CO.geometry(HW)
F1, F2, F3 = CO.FILTERS


def step_flops(B):
    r1, r3, r5 = B * g["c1"][0] * g["c1"][1], B * g["c3"][0] * g["c3"][1], B * g["c5"][0] * g["c5"][1]
    d11, d13, d15 = r5, B * g["u12"][0] * g["u12"][1], B * g["u14"][0] * g["u14"][1]
    conv = [(r1, F1, 25), (r3, F2, 25 * F1), (r5, F3, 9 * F2)]
    deconv = [(d11, F3, 9 * F2), (d13, F2, 25 * F1), (d15, F1, 25)]
    dense = [(B, 500, g["flat"]), (B, 50, 500), (B, 500, 50), (B, g["flat"], 500)]
    fwd = sum(2.0 * m * n * k for m, n, k in conv + deconv + dense)
    bwd = sum(2.0 * m * n * k * 2 for m, n, k in conv[1:] + deconv + dense) + 2.0 * conv[0][0] * conv[0][1] * conv[0][2]
    return fwd, fwd + bwd


rng = np.random.RandomState(0)
for B in (128, 1024, 4096):                # 128 = the reference's batch; the larger ones show where the kernels saturate
    x = torch.as_tensor(np.tanh(rng.normal(size=(B, HW[0] * HW[1]))).astype(np.float32), device="cuda")
    fwd_fl, step_fl = step_flops(B)
    print("batch %d frames of %dx%d; GEMM flops per step %.1f G (forward %.1f G = %.1f MFLOP/frame)" %
          (B, HW[0], HW[1], step_fl / 1e9, fwd_fl / 1e9, fwd_fl / B / 1e6))
    for prec, peak in (("f32", 157.3), ("bf16", 2500.0)):
        m = ConvAE(HW, 500, 50, prec)
        m.init_params(rng)
        for _ in range(3):
            m.train(x, want_loss=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            m.train(x, want_loss=False)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        for _ in range(3):
            m.encode(x)
        a.record()
        for _ in range(10):
            m.encode(x)
        b.record()
        torch.cuda.synchronize()
        ems = a.elapsed_time(b) / 10
        print("  %-5s train step %.3f ms = %.0f frames/s, %.1f TFLOP/s (%.1f %% of the %s MFMA peak); encoder alone %.3f ms = %.0f frames/s"
              % (prec, ms, B / ms * 1e3, step_fl / ms / 1e9, 100 * step_fl / ms / 1e9 / peak, prec, ems, B / ems * 1e3))
        m.close()
    del x
x = torch.as_tensor(np.tanh(rng.normal(size=(128, HW[0] * HW[1]))).astype(np.float32), device="cuda")
p = CO.init_params(np.random.default_rng(0), np.float32)
xs = x.cpu().numpy()[:16]
CO.loss_and_grads(p, xs)
t = time.perf_counter()
for _ in range(2):
    CO.loss_and_grads(p, xs)
cpu = (time.perf_counter() - t) / 2
print("NumPy oracle on the host (%d cores): %.2f s per 16-frame step = %.1f frames/s" % (os.cpu_count(), cpu, 16 / cpu))
