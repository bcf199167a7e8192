"""Probe: the B = 26 step time in deterministic mode after sub-runs of other arithmetics (bench.py's flow), with losses.
usage: python profiles/scripts/det_b26_probe.py <steps at B=520 per arithmetic> <list of arithmetics run before, e.g. bf16,f32>"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ip_avsr_amd.model import AdeNetModel
steps = int(sys.argv[1]); seq = sys.argv[2].split(",")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
m = AdeNetModel(bench.build_spec()); m.set_precision("bf16"); bench.synthetic_params(m)
xs, y, m_d, mask = bench.synthetic_batch(torch, 0, bench.B_PER_GPU, dev)
lens = np.asarray(mask).sum(axis=1).astype(np.int32)
xb, yb, mb_d, mask_b = bench.synthetic_batch(torch, 2000, 26, dev)
lens_b = np.asarray(mask_b).sum(axis=1).astype(np.int32)
if os.environ.get("PROBE_BF16_INPUTS"):
    xs = [x.to(torch.bfloat16) for x in xs]; xb = [x.to(torch.bfloat16) for x in xb]
def run(prec, n, B520=True):
    m.set_precision(prec)
    t0 = None
    for k in range(n + 3):
        if k == 3: torch.cuda.synchronize(); t0 = time.perf_counter()
        if B520:
            m.set_batch_lengths(lens); loss = m.train_step(xs, y, m_d, bench.THETA, bench.LR, want_loss=(k == n + 2))
        else:
            if os.environ.get("PROBE_ANNOUNCE"): m.set_batch_lengths(lens_b)
            loss = m.train_step(xb, yb, mb_d, bench.THETA, bench.LR, want_loss=(k == n + 2))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, loss
for prec in seq:
    ms, loss = run(prec, steps)
    print("%-6s B=520: %.3f ms/step, loss %.5f" % (prec, ms, loss), flush=True)
for rep in range(3):
    ms, loss = run("bf16", 20, B520=False)
    print("bf16   B=26 : %.3f ms/step, loss %.5f" % (ms, loss), flush=True)
