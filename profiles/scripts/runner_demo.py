"""The product entry point from the command line, end to end: writes the learnable AVLetters-shaped set of
tests/learnable_avletters.py (780 utterances, three 1200-pixel streams, the reference's schema-1 .ini + .mat files) to a scratch
directory and trains on it with `python ip_avsr_amd/runners/3stream.py --config ...` -- .mat loading, the reference's
preprocessing switches, splits uploaded once, minibatches gathered on the GPU, Adam, per-epoch evaluation -- printing the
driver's own epoch lines (the seconds in brackets are the reference's `time.time() - time_start`).

    python profiles/scripts/runner_demo.py [bf16|bf16x3|f32] [epochs]
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import learnable_avletters as LA

precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
root = tempfile.mkdtemp(prefix="avletters_demo_")
t0 = time.time()
ini = LA.build(root, seed=1234, amplitude=tuple(5.0 * a for a in (0.16, 0.12, 0.10)), num_epoch=epochs, validation_window=epochs)
print("dataset + DBN files written in %.1f s: %s" % (time.time() - t0, ini), flush=True)
# (a throw-away one-epoch run first: the first process of a fresh box pages in the ROCm libraries and torch, ~5 s)
subprocess.run([sys.executable, os.path.join(ROOT, "ip_avsr_amd", "runners", "3stream.py"), "--config", LA.build(tempfile.mkdtemp(prefix="avletters_warm_"),
                seed=1234, num_epoch=1, validation_window=1), "--seed", "1", "--precision", precision], cwd=ROOT, stdout=subprocess.DEVNULL,
               stderr=subprocess.DEVNULL)
for env_extra, label in (({}, "splits resident in HBM, minibatches gathered on the GPU (default)"),
                         ({"ADN_HOST_BATCHES": "1"}, "ADN_HOST_BATCHES=1: the reference's host-side batch assembly, upload per batch")):
    print("== %s, --precision %s" % (label, precision), flush=True)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "ip_avsr_amd", "runners", "3stream.py"), "--config", ini, "--seed", "1234",
                          "--precision", precision, "--write_results", os.path.join(root, "res.csv")],
                         env=dict(os.environ, **env_extra), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    lines = [l.split("\r")[-1] for l in out.stdout.splitlines()]
    for l in lines:
        if l.startswith("Epoch ") and "train cost" in l or l.startswith("CR:") or l.startswith("Final"):
            print("   " + l)
    print("   process wall time %.1f s (python start, .mat loading, model build, %d epochs, report); exit code %d"
          % (time.time() - t0, epochs, out.returncode), flush=True)
