"""Training-step throughput of every BASELINE.json configuration's shape, synthetic data resident in HBM,
at the reference's own minibatch and at the whole-split batch, in the f32, bf16x3 (fp32-grade on the bf16 matrix pipe) and bf16 modes.  One step = forward + temporal softmax loss +
BPTT + Adam on one batch.  These are parity-test shapes, not the bench line (bench.py is configs[1] at B = 520); the
table shows how the kernels hold up away from the tuned shape.

    python profiles/configs_bench.py            (on an MI355X)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ip_avsr_amd.modelzoo import (adenet_3stream, adenet_3stream_dct, adenet_4stream, adenet_v2, adenet_v2_1,
                                  deltanet_majority_vote)

rng = np.random.RandomState(1234)
T, THETA = 40, 9
SHP, MSK = lambda d: (None, None, d), (None, None)


def ae(din, acts=("rectify", "rectify", "rectify", "linear")):
    dims = [din, 2000, 1000, 500, 50]
    return ([(rng.normal(size=(a, b)) * 0.01).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])],
            [np.zeros(b, np.float32) for b in dims[1:]], dims[1:], list(acts))


CONFIGS = [
    # name, input dims, classes, batches, factory
    ("configs[0] AVLetters unimodal DeltaNet (deltanet_majority_vote, BLSTM-250)", (1200,), 26, (26, 520),
     lambda: deltanet_majority_vote.create_model(ae(1200), SHP(1200), None, MSK, None, 250, None, 26, 'glorot', False, True)),
    ("configs[1] AVLetters trimodal AdeNet, 3 encoder streams (adenet_3stream, concat) -- the bench line's model", (1200, 1200, 1200), 26, (26, 520),
     lambda: adenet_3stream.create_model(ae(1200), ae(1200), ae(1200), SHP(1200), None, SHP(1200), None, SHP(1200), None,
                                         MSK, None, 250, None, 26, 'concat', 'glorot', False)),
    ("configs[1] AVLetters trimodal AdeNet, raw + diff + DCT (adenet_3stream_dct, concat)", (1200, 1200, 90), 26, (26, 520),
     lambda: adenet_3stream_dct.create_model(ae(1200), ae(1200), SHP(1200), None, SHP(1200), None, SHP(90), None,
                                             MSK, None, 250, None, 26, 'concat', 'glorot', False)),
    ("configs[2] CUAVE bimodal (adenet_v2: 1500-d encoder + DCT stream, sum, 10 classes)", (1500, 90), 10, (10, 520),
     lambda: adenet_v2.create_model(ae(1500), SHP(1500), None, MSK, None, SHP(90), None, 250, None, 10, 'sum', 'glorot', False)),
    ("configs[3] OuluVS AdeNet-v2_1 (raw + diff 1144-d encoders, concat, peepholes, 10 classes)", (1144, 1144), 10, (10, 520),
     lambda: adenet_v2_1.create_model(ae(1144), ae(1144), SHP(1144), None, MSK, None, SHP(1144), None, 250, None, 10,
                                      'concat', 'ortho', True)),
    ("configs[4] AdeNet 4-stream, 512-unit LSTMs / BLSTM (adenet_4stream, concat)", (1200, 1200, 1200, 1200), 26, (26, 520),
     lambda: adenet_4stream.create_model(ae(1200), ae(1200), ae(1200), ae(1200), SHP(1200), None, SHP(1200), None, SHP(1200), None,
                                         SHP(1200), None, MSK, None, 512, None, 26, 'concat', 'glorot', False)),
]


def unwrap(m):
    return m[0] if isinstance(m, tuple) else m


print("%-100s %5s %12s %12s %12s" % ("configuration (T = %d, theta = %d)" % (T, THETA), "B", "f32 seq/s", "bf16x3 seq/s", "bf16 seq/s"))
for name, dims, classes, batches, make in CONFIGS:
    for B in batches:
        lens = rng.randint(12, T + 1, size=B); lens[0] = T
        mask = torch.as_tensor((np.arange(T)[None, :] < lens[:, None]).astype(np.uint8), device="cuda")
        x = [torch.as_tensor(rng.normal(size=(B, T, d)).astype(np.float32), device="cuda") * mask[..., None] for d in dims]
        y = torch.as_tensor(np.repeat(rng.randint(0, classes, size=(B, 1)), T, axis=1).astype(np.int32), device="cuda")
        rate = {}
        for precision in ("f32", "bf16x3", "bf16"):
            m = unwrap(make())
            m.set_precision(precision)
            for _ in range(20):                      # (the GPU idles while the host builds the model: at the small batches the
                m.train_step(x, y, mask, THETA, 1e-4, want_loss=False)      #  first ~10 steps can run 3x slower until the clocks are back)
            m.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            steps = 10
            a.record()
            for _ in range(steps):
                m.train_step(x, y, mask, THETA, 1e-4, want_loss=False)
            b.record()
            torch.cuda.synchronize()
            rate[precision] = B * steps / (a.elapsed_time(b) * 1e-3)
            m.close()
        print("%-100s %5d %12.0f %12.0f %12.0f" % (name, B, rate["f32"], rate["bf16x3"], rate["bf16"]))
