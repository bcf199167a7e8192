"""In-kernel phase stamps of the WIDE bf16x3 LSTM kernels (256 < H <= 512; library built with -DADN_LSTM_STAMPS:
profiles/scripts/build_alt.sh) on BASELINE configs[4] at B = 520: microseconds per time step and phase of one workgroup."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ip_avsr_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "profiles", "alt", "libadenet_hip.so")
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
m.set_precision("bf16x3")
lens = rng.randint(12, T + 1, size=B); lens[0] = T
mask = torch.as_tensor((np.arange(T)[None, :] < lens[:, None]).astype(np.uint8), device="cuda")
x = [torch.as_tensor(rng.normal(size=(B, T, 1200)).astype(np.float32), device="cuda") * mask[..., None] for _ in range(4)]
y = torch.as_tensor(np.repeat(rng.randint(0, 26, size=(B, 1)), T, axis=1).astype(np.int32), device="cuda")
lib = _lib.load()
for _ in range(3):
    m.train_step(x, y, mask, THETA, 1e-4, want_loss=False)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
lib.adn_debug_lstm_stamps(out, 1)
n = 5
for _ in range(n):
    m.train_step(x, y, mask, THETA, 1e-4, want_loss=False)
torch.cuda.synchronize()
lib.adn_debug_lstm_stamps(out, 0)
names = ["fwd product + handoff barrier", "fwd gate math + publish", "fwd outputs", "fwd poll + fill", "fwd barrier",
         "bwd product + sends", "bwd state request + collect", "bwd barrier + gate math + barrier"]
steps = n * 40 * 6.0          # one LSTM per launch at this size: all six carry blockIdx.y == 0
for k in range(8):
    print("%-36s %8.3f us per step" % (names[k], out[k] / 100.0 / steps))
