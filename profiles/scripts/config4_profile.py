"""10 train steps of BASELINE configs[4] (4-stream AdeNet, 512-unit LSTMs) at B = 520 for
`rocprofv3 --kernel-trace --stats -- python3 profiles/scripts/config4_profile.py [bf16x3|bf16|f32]`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ip_avsr_amd.modelzoo import adenet_4stream

rng = np.random.RandomState(1234)
T, THETA, B = 40, 9, int(os.environ.get("C4_BATCH", 520))
SHP, MSK = lambda d: (None, None, d), (None, None)


def ae(din):
    dims = [din, 2000, 1000, 500, 50]
    return ([(rng.normal(size=(a, b)) * 0.01).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])],
            [np.zeros(b, np.float32) for b in dims[1:]], dims[1:], ["rectify", "rectify", "rectify", "linear"])


m = adenet_4stream.create_model(ae(1200), ae(1200), ae(1200), ae(1200), SHP(1200), None, SHP(1200), None, SHP(1200), None, SHP(1200), None,
                                MSK, None, 512, None, 26, 'concat', 'glorot', False)
m = m[0] if isinstance(m, tuple) else m
m.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
lens = rng.randint(12, T + 1, size=B); lens[0] = T
mask = torch.as_tensor((np.arange(T)[None, :] < lens[:, None]).astype(np.uint8), device="cuda")
x = [torch.as_tensor(rng.normal(size=(B, T, 1200)).astype(np.float32), device="cuda") * mask[..., None] for _ in range(4)]
y = torch.as_tensor(np.repeat(rng.randint(0, 26, size=(B, 1)), T, axis=1).astype(np.int32), device="cuda")
import time
for it in range(int(os.environ.get("C4_STEPS", 10)) + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    m.train_step(x, y, mask, THETA, 1e-4, want_loss=False)
torch.cuda.synchronize()
print("configs[4] %s B=%d: %.3f ms per train step" % (sys.argv[1] if len(sys.argv) > 1 else "bf16x3", B, (time.perf_counter() - t0) * 1e3 / (it - 1)), flush=True)
