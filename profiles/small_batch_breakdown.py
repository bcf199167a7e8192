"""Per-kernel-class time of one training step at the reference's own minibatch sizes (B=26, adenet_v3 with 500-unit
LSTMs; B=10..26 elsewhere), from the library's HIP-event profiler.   python profiles/small_batch_breakdown.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ip_avsr_amd.modelzoo import adenet_v3

rng = np.random.RandomState(0)
B, T, THETA = int(os.environ.get("SB_BATCH", 26)), 40, 9


def encoder(din):
    dims = [din, 2000, 1000, 500, 50]
    return ([(rng.normal(size=(a, b)) * np.sqrt(2.0 / (a + b))).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])],
            [np.zeros(b, np.float32) for b in dims[1:]])


lens = rng.randint(12, T + 1, size=B); lens[0] = T
mask = torch.as_tensor((np.arange(T)[None, :] < lens[:, None]).astype(np.uint8), device="cuda")
x = [torch.as_tensor(rng.normal(size=(B, T, d)).astype(np.float32), device="cuda") for d in (1200, 90, 1200)]
y = torch.as_tensor(np.repeat(rng.randint(0, 26, size=(B, 1)), T, axis=1).astype(np.int32), device="cuda")
for precision in ("f32", "bf16"):
    m, _ = adenet_v3.create_model(encoder(1200), encoder(1200), (None, None, 1200), None, (None, None), None,
                                  (None, None, 90), None, (None, None, 1200), None, int(os.environ.get("SB_H", 250)), None, 26, "sum")
    m.set_precision(precision)
    for _ in range(3):
        m.compute_grads(x, y, mask, THETA, want_loss=False); m.apply_adadelta(2.0)
    m.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        m.compute_grads(x, y, mask, THETA, want_loss=False); m.apply_adadelta(2.0)
    b.record(); torch.cuda.synchronize()
    print("%s: B=%d step %.3f ms" % (precision, B, a.elapsed_time(b) / 20))
    m.profile(True)
    for _ in range(5):
        m.compute_grads(x, y, mask, THETA, want_loss=False); m.apply_adadelta(2.0)
    m.synchronize()
    prof = m.profile_read()
    m.profile(False)
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"] if isinstance(kv[1], dict) else 0):
        print("   ", k, v)
    m.close()
