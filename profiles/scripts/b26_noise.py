"""Run-to-run and grouped-vs-single differences of the gradients at B = 26 (bf16 mode): is a 1e-3 relative difference summation
noise (split-K float atomics in arrival order) or a property of the grouped launches?"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_bench_geometry as G
d = tempfile.mkdtemp()
base = G._run(d, "default")
ref = os.path.join(d, "default.npz")
runs = {"a": G._run(d, "a", ref, GEOM_BATCH="26"), "b": G._run(d, "b", ref, GEOM_BATCH="26"),
        "single_a": G._run(d, "sa", ref, GEOM_BATCH="26", ADN_GEMM_NO_RS_GROUPS="1"),
        "single_b": G._run(d, "sb", ref, GEOM_BATCH="26", ADN_GEMM_NO_RS_GROUPS="1")}
def rel(x, y, k):
    a, b = x[k].astype(np.float64).ravel(), y[k].astype(np.float64).ravel()
    return np.linalg.norm(a - b) / np.linalg.norm(b)
for k in ("g_fc1_s1.W", "g_bottleneck_s3.W", "g_lstm_s3.W_in_to_ingate", "g_lstm_s1.W_hid_to_cell"):
    print(k, "grouped twice %.1e | single twice %.1e | grouped vs single %.1e" % (rel(runs["a"], runs["b"], k), rel(runs["single_a"], runs["single_b"], k), rel(runs["a"], runs["single_a"], k)))
