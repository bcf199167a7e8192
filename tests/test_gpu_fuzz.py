"""Randomised configurations through the whole path (tests/fuzz_model.py: f32 mode vs the oracle over random graphs;
tests/fuzz_lstm.py: the resident-weight bf16 LSTM kernels vs the one-workgroup kernels over random shapes)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script)] + list(args), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:]
    return res.stdout


def test_random_graphs_match_the_oracle_in_f32():
    out = run("fuzz_model.py", "24", "2025")
    assert "bad: 0" in out, "\n".join(l for l in out.splitlines() if " BAD " in l or "bad:" in l)


def test_random_graphs_track_the_oracle_in_bf16():
    out = run("fuzz_model.py", "24", "2026", "bf16")
    assert "bad: 0" in out, "\n".join(l for l in out.splitlines() if " BAD " in l or "bad:" in l)


def test_random_graphs_match_the_oracle_in_bf16x3():
    """Every GEMM of the random graphs -- K from 2 up, ragged M and N -- through the hi / lo split path."""
    out = run("fuzz_model.py", "24", "2027", "bf16x3")
    assert "bad: 0" in out, "\n".join(l for l in out.splitlines() if " BAD " in l or "bad:" in l)


def test_random_lstm_shapes_cluster_equals_single_workgroup():
    out = run("fuzz_lstm.py")
    assert "bad: 0" in out, "\n".join(l for l in out.splitlines() if "BAD" in l or "NON-REPEATABLE" in l or "bad:" in l)
