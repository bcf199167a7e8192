import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ip_avsr_amd.modelzoo import adenet_v2_1, adenet_v2
rng = np.random.RandomState(1234)
T, THETA = 40, 9
SHP, MSK = lambda d: (None, None, d), (None, None)
def ae(din, acts=("rectify", "rectify", "rectify", "linear")):
    dims = [din, 2000, 1000, 500, 50]
    return ([(rng.normal(size=(a, b)) * 0.01).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])],
            [np.zeros(b, np.float32) for b in dims[1:]], dims[1:], list(acts))
for peep in (True, False):
  for init in ("ortho", "glorot"):
    m = adenet_v2_1.create_model(ae(1144), ae(1144), SHP(1144), None, MSK, None, SHP(1144), None, 250, None, 10, 'concat', init, peep)
    m = m[0] if isinstance(m, tuple) else m
    B = 10
    lens = rng.randint(12, T + 1, size=B); lens[0] = T
    mask = torch.as_tensor((np.arange(T)[None, :] < lens[:, None]).astype(np.uint8), device="cuda")
    x = [torch.as_tensor(rng.normal(size=(B, T, d)).astype(np.float32), device="cuda") * mask[..., None] for d in (1144, 1144)]
    y = torch.as_tensor(np.repeat(rng.randint(0, 10, size=(B, 1)), T, axis=1).astype(np.int32), device="cuda")
    m.set_precision("f32")
    for _ in range(3): m.train_step(x, y, mask, THETA, 1e-4, want_loss=False)
    m.synchronize()
    m.profile(True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): m.train_step(x, y, mask, THETA, 1e-4, want_loss=False)
    b.record(); torch.cuda.synchronize()
    pr = m.profile_read()
    print("peep", peep, init, "ms/step %.2f" % (a.elapsed_time(b) / 10), {k: round(v["ms"] / 10, 3) for k, v in pr.items()})
    m.profile(False); m.close()
