"""Builds ip_avsr_amd/csrc/libadenet_hip.so in-tree with hipcc for gfx950.

    python -m ip_avsr_amd.build [--force]

hipcc cross-compiles without a GPU, so this also runs in a CPU-only container.
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libadenet_hip.so")


def build(force=False, verbose=True):
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    cmd = ["make", "-C", CSRC, "-j", str(min(4, os.cpu_count() or 1))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode:
        sys.stdout.write(res.stdout)
    if res.returncode:
        raise RuntimeError("building libadenet_hip.so failed (exit %d)" % res.returncode)
    if not os.path.exists(LIB):
        raise RuntimeError("make succeeded but %s is missing" % LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
