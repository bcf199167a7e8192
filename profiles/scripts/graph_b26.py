"""Would a captured HIP graph shorten the train step at the reference's own minibatch (B = 26)?  A MEASUREMENT of the launch
side only: one bf16 train step of the bench model is captured with torch.cuda.graph and replayed.  The replay is NOT a
valid training loop -- the kernel arguments that change per step (Adam's bias correction, the LSTM exchange's launch tag,
the dropout counter) are frozen in the graph -- so its results are discarded; its TIME is what launching the same ~70
kernels with zero host work costs.  Printed beside the eager loop (one C call per step, no host synchronisation)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ip_avsr_amd.model import AdeNetModel  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
for B in (26, 520):
    m = AdeNetModel(bench.build_spec())
    m.set_precision("bf16")
    bench.synthetic_params(m)
    xs, y, m_d, _ = bench.synthetic_batch(torch, 0, B, dev)
    xs = [x.to(torch.bfloat16) for x in xs]
    step = lambda: m.train_step(xs, y, m_d, bench.THETA, 1e-3, want_loss=False)
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    n = 200 if B == 26 else 50
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            step()
        torch.cuda.synchronize()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n
        print("B = %3d: eager %.3f ms / step, graph replay %.3f ms / step (%.1f %%)" % (B, 1e3 * eager, 1e3 * graph,
                                                                                       100.0 * (graph / eager - 1.0)))
    except Exception as e:                            # a call that cannot be captured
        print("B = %3d: eager %.3f ms / step; capture failed: %s" % (B, 1e3 * eager, str(e)[:200]))
    try:
        m.close()
    except Exception:
        pass
