"""Feature front-end: each GPU transform timed at the AVLetters scale (780 utterances, 12-40 frames,
30x40 pixels -> ~20k frames x 1200) with inputs resident in HBM, beside this package's NumPy port on the host cores.
Algorithmic bytes = what the transform must read and write once (fp32).

    python profiles/prep_bench.py        (on an MI355X)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ip_avsr_amd.utils import preprocessing as host
from ip_avsr_amd.utils import preprocessing_gpu as gpu
from ip_avsr_amd.utils import lcn as gpu_lcn
from oracle import lcn_oracle                      # CPU leg only

rng = np.random.RandomState(1234)
lens = rng.randint(12, 41, size=780)
n = int(lens.sum())
X = rng.normal(size=(n, 1200)).astype(np.float32)
F30 = rng.normal(size=(n, 30)).astype(np.float32)
Xd, Fd = torch.as_tensor(X, device="cuda"), torch.as_tensor(F30, device="cuda")
e = 4.0


def gpu_ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def cpu_ms(fn, iters=3):
    fn()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t) / iters * 1e3


LCN = gpu_lcn.make_lecun_lcn((n, 1, 30, 40), (30, 40), 9)
cases = [
    ("compute_diff_images", lambda: gpu.compute_diff_images(Xd, lens), lambda: host.compute_diff_images(X, lens), 2 * e * n * 1200),
    ("sequencewise_mean_image_subtraction", lambda: gpu.sequencewise_mean_image_subtraction(Xd, lens),
     lambda: host.sequencewise_mean_image_subtraction(X, lens), 2 * e * n * 1200),
    ("normalize_input (in place)", lambda: gpu.normalize_input(Xd), lambda: host.normalize_input(X.copy()), 2 * e * n * 1200),
    ("featurewise_normalize_sequence", lambda: gpu.featurewise_normalize_sequence(Xd), lambda: host.featurewise_normalize_sequence(X),
     2 * e * n * 1200),
    ("reorder_data f->c", lambda: gpu.reorder_data(Xd, (30, 40)), lambda: host.reorder_data(X, (30, 40)), 2 * e * n * 1200),
    ("compute_dct_features zigzag 30", lambda: gpu.compute_dct_features(Xd, (30, 40), 30),
     lambda: host.compute_dct_features(X, (30, 40), 30), e * n * (1200 + 30)),
    ("concat_first_second_deltas F=30 w=9", lambda: gpu.concat_first_second_deltas(Fd, lens, 9),
     lambda: host.concat_first_second_deltas(F30, lens, 9), e * n * (30 + 90)),
    ("lecun_lcn 30x40, 9x9 (utils/lcn.py)", lambda: LCN(Xd), lambda: lcn_oracle.lecun_lcn(X[:2000], (30, 40), 9, dtype=np.float32),
     2 * e * n * 1200),
]
print("frames %d x 1200 (%.1f MB fp32); host cores %d" % (n, n * 1200 * 4 / 1e6, os.cpu_count()))
print("%-40s %10s %12s %12s %9s" % ("transform", "GPU ms", "GB/s (alg.)", "host ms", "ratio"))
for name, g, h, nbytes in cases:
    gm, hm = gpu_ms(g), cpu_ms(h)
    if name.startswith("lecun_lcn"):
        hm *= n / 2000.0                                # the host leg ran on 2000 frames
    print("%-40s %10.3f %12.0f %12.1f %8.0fx" % (name, gm, nbytes / gm / 1e6, hm, hm / gm))
