"""Trains the learnable synthetic AVLetters set (tests/learnable_avletters.py) with the package's 3-stream runner in each
arithmetic mode from the same seed(s) and prints the per-epoch curves side by side (the numbers tests/test_gpu_accuracy.py
asserts on).   python3 profiles/scripts/accuracy_explore.py [amplitude-scale] [epochs] [lr] [arms] [seeds]"""
import contextlib
import io
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import learnable_avletters as LA  # noqa: E402
from ip_avsr_amd.runners import nstream  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
arms = sys.argv[4].split(",") if len(sys.argv) > 4 else ["f32", "bf16x3", "bf16"]
seeds = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [1234]


def one_seed(ini, seed, final):
    res = {}
    print("---- seed %d" % seed)
    for arm in arms:
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            out = nstream.main(3, ["--config", ini, "--seed", str(seed), "--precision", arm])
        net, h = out["network"], out["heldout"]
        probs = net.predict(h["X_val"], h["mask_val"], out["windowsize"])
        lens = h["mask_val"].sum(-1)
        votes = np.array([np.bincount(probs[i, :lens[i]].argmax(-1), minlength=26).argmax() for i in range(len(probs))])
        res[arm] = dict(out=out, votes=votes, probs=probs)
        final[arm].append(float((votes == h["y_val"]).mean()))
        print("%-7s %.1f s  val cost %s" % (arm, time.time() - t0, " ".join("%.4f" % v for v in out["cost_val"])))
        print("        class rate %s  final %.4f test %.4f" % (" ".join("%.3f" % v for v in out["class_rate"]), final[arm][-1],
                                                               out["test_cr"]))
        net.close()
    ref = res[arms[0]]
    for arm in arms[1:]:
        r = res[arm]
        n = min(len(r["out"]["cost_val"]), len(ref["out"]["cost_val"]))
        dv = np.abs(np.array(r["out"]["cost_val"][:n]) - np.array(ref["out"]["cost_val"][:n]))
        print("%s vs %s: votes differing %d / %d, max |dp| %.3e, val-cost curve max |d| %.3e (rel %.3e)"
              % (arm, arms[0], int((r["votes"] != ref["votes"]).sum()), len(ref["votes"]), np.abs(r["probs"] - ref["probs"]).max(),
                 dv.max(), (dv / np.array(ref["out"]["cost_val"][:n])).max()))


with tempfile.TemporaryDirectory(dir="/tmp") as root:
    t0 = time.time()
    ini = LA.build(root, amplitude=tuple(scale * a for a in (0.16, 0.12, 0.10)), num_epoch=epochs, learning_rate=lr,
                   validation_window=epochs)
    print("dataset written in %.1f s" % (time.time() - t0))
    final = {a: [] for a in arms}
    for seed in seeds:
        one_seed(ini, seed, final)
    for a in arms:
        print("final class rate, %-7s mean %.4f  std %.4f  %s" % (a, np.mean(final[a]), np.std(final[a]), final[a]))
