"""Where does the ping-pong kernel's forward output differ from the register-staged kernels'?  (debug aid)
python profiles/scripts/pp_bitdiff.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RUN = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import bench
from ip_avsr_amd.model import AdeNetModel
torch.cuda.set_device(0)
m = AdeNetModel(bench.build_spec()); m.set_precision("bf16"); bench.synthetic_params(m)
xs, y, m_d, mask = bench.synthetic_batch(torch, 0, 520, torch.device("cuda", 0))
m.predict(xs, m_d, bench.THETA)
out = {}
for l in range(4):
    out["a%%d" %% l] = m.encoder_activation(0, l, 520, 40)
np.savez(sys.argv[1], **out)
''' % ROOT
res = {}
for tag, env in (("pp", {}), ("nopp", {"ADN_GEMM_PP": "0"})):
    f = "/tmp/bitdiff_%s.npz" % tag
    subprocess.run([sys.executable, "-c", RUN, f], check=True, env=dict(os.environ, **env))
    res[tag] = dict(np.load(f))
for l in range(4):
    a, b = res["pp"]["a%d" % l], res["nopp"]["a%d" % l]
    d = a != b
    print("layer", l, a.shape, "mismatched", int(d.sum()), "max abs", float(np.abs(a - b).max()))
    if d.any():
        rows, cols = np.nonzero(d)
        print("  rows mod 16 histogram", np.bincount(rows % 16, minlength=16).tolist())
        print("  cols mod 32 histogram", np.bincount(cols % 32, minlength=32).tolist())
        print("  cols // 16 mod 8 histogram", np.bincount((cols // 16) % 8, minlength=8).tolist())
        print("  rows // 16 mod 4 histogram", np.bincount((rows // 16) % 4, minlength=4).tolist())
        print("  first few:", [(int(r), int(c), float(a[r, c]), float(b[r, c])) for r, c in list(zip(rows, cols))[:6]])
        break
