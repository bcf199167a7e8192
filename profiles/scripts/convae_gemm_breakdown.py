"""Per-launch GEMM table of one conv auto-encoder train step (batch CAE_BATCH, default 1024, bf16):
    ADN_GEMM_TRACE=1 rocprofv3 --kernel-trace -d <dir> -o bd --output-format csv -- python3 profiles/scripts/convae_gemm_breakdown.py run 2> trace.txt
    python3 profiles/scripts/convae_gemm_breakdown.py join trace.txt <dir>/.../bd_kernel_trace.csv"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))

if sys.argv[1] == "run":
    import numpy as np
    import torch
    from ip_avsr_amd.convae import ConvAE
    m = ConvAE((30, 40), 500, 50, os.environ.get("CAE_PRECISION", "bf16"))
    m.init_params(np.random.RandomState(0))
    x = torch.as_tensor(np.tanh(np.random.RandomState(1).normal(size=(int(os.environ.get("CAE_BATCH", 1024)), 1200))).astype(np.float32), device="cuda")
    m.train(x, want_loss=False)
    torch.cuda.synchronize()
    for _ in range(2):
        sys.stderr.write("ADN_STEP\n"); sys.stderr.flush()
        m.train(x, want_loss=False)
        torch.cuda.synchronize()
else:
    import gemm_breakdown
    gemm_breakdown.join(sys.argv[2], sys.argv[3])
