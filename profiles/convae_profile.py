"""10 conv-AE train steps (batch 128) for `rocprofv3 --kernel-trace --stats -- python3 profiles/convae_profile.py [f32|bf16]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ip_avsr_amd.convae import ConvAE

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
m = ConvAE((30, 40), 500, 50, prec)
m.init_params(np.random.RandomState(0))
x = torch.as_tensor(np.tanh(np.random.RandomState(1).normal(size=(int(os.environ.get("CAE_BATCH", 128)), 1200))).astype(np.float32), device="cuda")
for _ in range(10):
    m.train(x, want_loss=False)
torch.cuda.synchronize()
