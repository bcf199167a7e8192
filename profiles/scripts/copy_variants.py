"""The float4 copy yardstick of bench.py (adn_op_copy_bench) in its variants: which form reaches the guide's 6.29 TB/s?
   python profiles/scripts/copy_variants.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = "import sys; sys.path.insert(0, %r); import torch, bench; print('%%.0f' %% bench.measured_hbm_gbs(torch, torch.device('cuda', 0)))" % ROOT
for blocks in (8, 16, 32, 64):
    for variant in (0, 2, 4, 8, 18, 20, 24):
        env = dict(os.environ, ADN_COPY_VARIANT=str(variant), ADN_COPY_BLOCKS=str(blocks))
        out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip().splitlines()
        print("blocks per CU %3d  variant %2d (unroll %d%s): %s GB/s" % (blocks, variant, variant & 15, ", non-temporal stores" if variant & 16 else "", out[-1] if out else "failed"))
