"""In-kernel phase stamps of the bf16x3 forward LSTM kernel (library built with -DADN_LSTM_STAMPS: profiles/scripts/build_alt.sh).
Prints 100 MHz ticks per phase, summed over the steps of one workgroup, for a few train steps at the bench geometry."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ip_avsr_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "profiles", "alt", "libadenet_hip.so")
import torch
import bench
from ip_avsr_amd.model import AdeNetModel

m = AdeNetModel(bench.build_spec())
m.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
bench.synthetic_params(m)
xs, y, m_d, _ = bench.synthetic_batch(torch, 0, bench.B_PER_GPU, torch.device("cuda", 0))
lib = _lib.load()
for _ in range(5):
    m.train_step(xs, y, m_d, bench.THETA, bench.LR, want_loss=False)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
lib.adn_debug_lstm_stamps(out, 1)
n = 10
for _ in range(n):
    m.train_step(xs, y, m_d, bench.THETA, bench.LR, want_loss=False)
torch.cuda.synchronize()
lib.adn_debug_lstm_stamps(out, 0)
names = ["product (+ handoff barrier)", "gate math + publish", "barrier + outputs + own k-steps", "poll", "fill + barrier", "(bwd) product+publish", "(bwd) poll", "(bwd) gate math"]
steps = n * 40 * 2.0          # two launches per train step carry blockIdx.y == 0 (stream LSTMs, aggregation pair)
for k in range(8):
    print("%-34s %8.3f us per step" % (names[k], out[k] / 100.0 / steps))
