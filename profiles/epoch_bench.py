"""The one workload the reference records a wall time for: an AVLetters trimodal `adenet_v3` epoch
(avletters/trimodal.py:356-420 -- 20 minibatches of 26 utterances trained with adadelta and dropout, then the train cost
on the last minibatch, the validation cost and the predictions on all 260 test utterances; Theta=9, H=250 -> LSTMs of
500 units, fusion 'sum').  The notebook logs 103.7-114.6 s per epoch on hardware it does not name
(avletters/avletters_training.ipynb:688-707), so that figure is context, not a baseline.

Inputs start as host NumPy arrays every call, as in the reference loop: the times below include the PCIe copies and the
host-side batch assembly.  Synthetic data of the AVLetters shape: lengths UniformInt[12,40], 520/260.

    python profiles/epoch_bench.py [--cpu-steps 2]       (on an MI355X)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ip_avsr_amd.modelzoo import adenet_v3

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=5)
ap.add_argument("--cpu-steps", type=int, default=2)
args = ap.parse_args()

EPOCH_SIZE, BATCH, THETA, CLASSES, TMAX = 20, 26, 9, 26, 40
rng = np.random.RandomState(1234)


def encoder(din):
    dims = [din, 2000, 1000, 500, 50]
    ws = [(rng.normal(size=(a, b)) * np.sqrt(2.0 / (a + b))).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])]
    return ws, [np.zeros(b, np.float32) for b in dims[1:]]


def split(n):
    lens = rng.randint(12, TMAX + 1, size=n)
    mask = (np.arange(TMAX)[None, :] < lens[:, None]).astype(np.uint8)
    x = [(rng.normal(size=(n, TMAX, d)) * mask[..., None]).astype(np.float32) for d in (1200, 90, 1200)]
    y = np.repeat(rng.randint(0, CLASSES, size=(n, 1)), TMAX, axis=1).astype(np.int32)
    return x, y, mask


train_x, train_y, train_m = split(520)
val_x, val_y, val_m = split(260)


def minibatch():
    idx = rng.choice(520, BATCH, replace=False)
    return [x[idx] for x in train_x], train_y[idx], train_m[idx]


def epoch(model, precision_label=None):
    for _ in range(EPOCH_SIZE):
        x, y, m = minibatch()
        model.compute_grads(x, y, m, THETA, want_loss=True)
        model.apply_adadelta(2.0)
    cost = model.loss(x, y, m, THETA, deterministic=False)
    val_cost = model.loss(val_x, val_y, val_m, THETA)
    pred = model.predict(val_x, val_m, THETA)
    return cost, val_cost, float((pred.argmax(-1) == val_y[:, 0]).mean())


for precision in ("f32", "bf16"):
    model, _fuse = adenet_v3.create_model(encoder(1200), encoder(1200), (None, None, 1200), None, (None, None), None,
                                   (None, None, 90), None, (None, None, 1200), None, 250, None, CLASSES, "sum")
    model.set_precision(precision)
    epoch(model)
    model.synchronize()
    times = []
    for _ in range(args.epochs):
        t = time.perf_counter()
        cost, val_cost, cr = epoch(model)
        model.synchronize()
        times.append(time.perf_counter() - t)
    print("%-5s epoch (20 x 26 train + cost + 2 x 260 eval, host inputs): median %.1f ms, min %.1f ms; "
          "train cost %.3f val cost %.3f class rate %.3f" % (precision, 1e3 * np.median(times), 1e3 * min(times), cost, val_cost, cr))
    model.close()
print("reference notebook, same loop, unnamed hardware: 103.7-114.6 s per epoch (avletters_training.ipynb:688-707)")

if args.cpu_steps > 0:                               # the CPU restatement of one minibatch step, for scale
    from oracle import adenet_oracle as O
    spec = O.spec_adenet_v3(1200, 90, 1200, enc_acts=("sigmoid", "sigmoid", "sigmoid", "linear"), fusion="sum")
    p = O.init_params(spec, np.random.default_rng(0), np.float32)
    x, y, m = minibatch()
    dr = dict(seed=1, counter=0)
    O.loss_and_grads(spec, p, x, y, m, THETA, dropout=dr)
    t = time.perf_counter()
    for _ in range(args.cpu_steps):
        O.loss_and_grads(spec, p, x, y, m, THETA, dropout=dr)
    cpu = (time.perf_counter() - t) / args.cpu_steps
    print("NumPy oracle on the host (%d cores): %.2f s per 26-utterance forward+backward -> >= %.0f s per epoch"
          % (os.cpu_count(), cpu, cpu * EPOCH_SIZE))
