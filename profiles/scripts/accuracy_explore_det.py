import contextlib, io, os, sys, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np
from tests import learnable_avletters as LA
from ip_avsr_amd import _lib
from ip_avsr_amd.runners import nstream
_lib.load().adn_set_deterministic(1)
root = tempfile.mkdtemp()
ini = LA.build(root, seed=1234, amplitude=tuple(3.0 * a for a in (0.16, 0.12, 0.10)), num_epoch=60, validation_window=60)
for arm in ("f32", "bf16x3", "bf16"):
    for seed in (1, 2, 3, 4):
        with contextlib.redirect_stdout(io.StringIO()):
            out = nstream.main(3, ["--config", ini, "--seed", str(seed), "--precision", arm])
        cr = out["class_rate"]
        print(arm, seed, " ".join("%.3f" % cr[e] for e in (9, 19, 24, 29, 34, 39, 44, 49, 54, 59)), "val", "%.4f" % out["cost_val"][-1], flush=True)
        out["network"].close()
