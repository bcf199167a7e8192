"""Epoch wall time through the PRODUCT entry point: ip_avsr_amd/runners/nstream.fit -- the epoch loop of
reference runners/3stream.py:322-427 -- on the bench model (AVLetters trimodal AdeNet, BASELINE configs[1]) and synthetic splits
of the AVLetters shape (520 train / 260 val / 260 test utterances, lengths UniformInt[12, 40], 3 x 1200 features).

One epoch = 20 minibatches of 26 utterances (Adam), the train cost of the last minibatch, the validation cost, the
majority-vote evaluation of the 260 validation utterances and -- when the validation cost improved -- of the 260 test
utterances plus the "best parameters" snapshot: everything the reference loop does between two `Epoch ...` lines.

Arms: minibatches assembled on the GPU from HBM-resident splits (default; with and without the side-stream prefetch) against
the reference's host-side assembly with an upload per batch (ADN_HOST_BATCHES=1: what every driver did up to round 3), in the
three arithmetics; then the whole-train batch (B = 520) as the runner steps it, to set beside bench.py's ms_per_step.

    python profiles/epoch_bench.py [--epochs 8]       (on an MI355X)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench as B
from ip_avsr_amd.model import AdeNetModel
from ip_avsr_amd.runners import nstream

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=8)
ap.add_argument("--precisions", default="bf16,bf16x3,f32")
args = ap.parse_args()

torch.cuda.set_device(0)
device = torch.device("cuda", 0)
split_dev, ys, lens = B.synthetic_splits(torch, device)
split_host = {k: [x.cpu().numpy() for x in v] for k, v in split_dev.items()}
quiet = lambda *a, **k: None


def run(model, host, prefetch, epochsize, batchsize, epochs):
    np.random.seed(3)
    return nstream.fit(model, split_host if host else split_dev, ys, lens, 3, windowsize=B.THETA, num_epoch=epochs, epochsize=epochsize,
                       batchsize=batchsize, validation_window=1000, learning_rate=B.LR, say=quiet, progress=False,
                       host_batches=host, prefetch=prefetch)


for precision in args.precisions.split(","):
    model = AdeNetModel(B.build_spec())
    model.set_precision(precision)
    model.spec["precision"] = precision
    B.synthetic_params(model)
    for label, host, prefetch in (("HBM-resident splits, gather on a side stream", False, True),
                                  ("HBM-resident splits, gather on the model's stream", False, False),
                                  ("host assembly + upload per batch (round 3's drivers)", True, False)):
        st = run(model, host, prefetch, 20, 26, args.epochs)
        ep, tr = np.array(st["epoch_seconds"][2:]) * 1e3, np.array(st["train_seconds"][2:]) * 1e3
        print("%-6s epoch 20 x 26 + costs + votes, %-52s median %6.1f ms (min %6.1f), minibatch loop %6.1f ms = %.3f ms / step; "
              "resident dtype %s" % (precision, label + ":", np.median(ep), ep.min(), np.median(tr), np.median(tr) / 20,
                                     "host" if host else nstream.resident_dtype(model)))
    for label, host, prefetch in (("HBM-resident, side-stream gather", False, True), ("HBM-resident, in-stream gather", False, False),
                                  ("host assembly", True, False)):
        st = run(model, host, prefetch, 10, 520, 5)
        tr = np.array(st["train_seconds"][1:]) * 1e3 / 10
        print("%-6s runner step at B = 520, %-34s %.3f ms / step (median of %d x 10 steps)" % (precision, label + ":", np.median(tr), len(tr)))
    b = B.synthetic_batch(torch, 0, 520, device)
    xs = [x.to(torch.bfloat16) for x in b[0]] if nstream.resident_dtype(model) == "bfloat16" else b[0]
    for _ in range(10):
        model.train_step(xs, b[1], b[2], B.THETA, B.LR, want_loss=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        model.train_step(xs, b[1], b[2], B.THETA, B.LR, want_loss=False)
    torch.cuda.synchronize()
    print("%-6s bare train_step at B = 520, batch resident: %.3f ms / step" % (precision, (time.perf_counter() - t) * 50))
    del b, xs
    # the bare kernel time of the same work, for the gap: 20 steps at B = 26 with the batch already assembled
    b = B.synthetic_batch(torch, 2000, 26, device)
    xs = [x.to(torch.bfloat16) for x in b[0]] if nstream.resident_dtype(model) == "bfloat16" else b[0]
    for _ in range(30):
        model.train_step(xs, b[1], b[2], B.THETA, B.LR, want_loss=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100):
        model.train_step(xs, b[1], b[2], B.THETA, B.LR, want_loss=False)
    torch.cuda.synchronize()
    print("%-6s bare train_step at B = 26, batch resident: %.3f ms / step" % (precision, (time.perf_counter() - t) * 10))
    model.close()
print("reference notebook, adenet_v3 epoch of the same sizes, unnamed hardware: 103.7-114.6 s (avletters_training.ipynb:688-707)")
