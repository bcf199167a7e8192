"""ADN_PRECISION_BF16X3: every large GEMM as three bf16 MFMA products of the operands' bf16 hi / lo parts
(a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate) -- fp32-grade results at the bf16 matrix rate.

  * operator level: against an fp64 product, every layout, odd sizes, K up to 20800, bias / activation / accumulate;
  * model level: the north-star parity gate of the fp32 mode (encoder activations and probabilities <= 1e-4, identical
    majority votes, gradients 2e-4 of scale against the fp64 oracle) at the real AVLetters widths, and a three-step Adam
    trajectory against the oracle on a small graph."""
import ctypes as C

import numpy as np
import pytest

from oracle import adenet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.fixture(scope="module")
def lib():
    from ip_avsr_amd import _lib
    l = _lib.load()
    assert l.adn_device_count() >= 1
    return l


def dptr(t):
    return C.c_void_p(t.data_ptr())


def ragged_mask(rng, B, T):
    lens = rng.integers(min(T, max(2, T // 3)), T + 1, size=B)
    lens[0] = T
    m = np.zeros((B, T), np.uint8)
    for i, l in enumerate(lens):
        m[i, :l] = 1
    return m


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(130, 250, 150), (1040, 2000, 1200), (1043, 1998, 1203), (500, 52, 20800), (2000, 1000, 20800),
                                   (20800, 500, 1000), (20800, 500, 50), (20800, 1000, 150), (5000, 300, 44)])
def test_gemm_bf16x3_is_fp32_grade(torch_cuda, lib, layout, M, N, K):
    torch = torch_cuda
    from ip_avsr_amd import _lib as L
    rng = np.random.default_rng(M + N + K + layout)
    A = (rng.normal(size=(M, K)) * rng.uniform(0.1, 3.0, size=(1, K))).astype(np.float32)   # uneven scales along k
    Bm = rng.normal(size=(K, N)).astype(np.float32)
    a_d, b_d = torch.tensor(A, device="cuda"), torch.tensor(Bm, device="cuda")
    ref = (a_d.double() @ b_d.double()).cpu().numpy()
    pad = lambda n: (n + 7) // 8 * 8

    def dev(x):
        buf = torch.full((x.shape[0], pad(x.shape[1])), float("nan"), device="cuda")     # pads must never be read as data
        buf[:, :x.shape[1]] = x
        return buf
    ah, bh = (a_d, b_d) if layout == 0 else (a_d, b_d.T.contiguous()) if layout == 1 else (a_d.T.contiguous(), b_d)
    ap, bp = dev(ah), dev(bh)
    bias = torch.tensor(rng.normal(size=pad(N)).astype(np.float32), device="cuda")
    c_d = torch.full((M, pad(N)), 7.0, device="cuda")
    L.check(lib.adn_op_gemm_ex(layout, M, N, K, dptr(ap), ap.shape[1], dptr(bp), bp.shape[1], dptr(c_d), c_d.shape[1],
                               dptr(bias), 0, 0, L.PRECISION["bf16x3"], None))
    torch.cuda.synchronize()
    out = c_d.cpu().numpy()
    scale = np.sqrt(K) * 3.0                              # typical |sum| of K products of this data
    want = ref + bias.cpu().numpy()[None, :N]
    err = np.abs(out[:, :N] - want).max()
    # 2^-17 per product, random signs: ~sqrt(K) 7.6e-6 |a||b| -- the same order as fp32 accumulation's own rounding over K
    assert err <= 4e-5 * scale, (err, scale)              # bf16 alone would be ~4e-3 * scale
    assert (out[:, N:] == 7.0).all()
    # accumulate + rectify through the same path
    c2 = torch.tensor(np.ascontiguousarray(out), device="cuda")
    L.check(lib.adn_op_gemm_ex(layout, M, N, K, dptr(ap), ap.shape[1], dptr(bp), bp.shape[1], dptr(c2), c2.shape[1],
                               None, 0, 1, L.PRECISION["bf16x3"], None))
    torch.cuda.synchronize()
    assert np.abs(c2.cpu().numpy()[:, :N] - (want + ref)).max() <= 8e-5 * scale


def test_real_dimensions_parity_gate_in_bf16x3(torch_cuda, lib):
    """tests/test_gpu_parity.py::test_real_dimensions_encoder_1e4_and_top1 with the GEMMs on the bf16 matrix pipe."""
    from ip_avsr_amd.model import AdeNetModel
    spec = O.spec_nstream([1200, 1200, 1200])
    B, T, theta = 10, 14, 9
    rng = np.random.default_rng(1234)
    p = O.init_params(spec, rng, np.float32, enc_std=0.01)
    for k in p:
        if k.endswith(".b"):
            p[k] = rng.normal(0, 0.05, p[k].shape).astype(np.float32)
    mask = ragged_mask(rng, B, T)
    inputs = [(rng.normal(size=(B, T, 1200)) * mask[..., None]).astype(np.float32) for _ in range(3)]
    y = np.repeat((np.arange(B) % 26)[:, None], T, axis=1).astype(np.int32)
    m = AdeNetModel(spec)
    m.set_precision("bf16x3")
    m.set_params_dict(p)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    x64 = [x.astype(np.float64) for x in inputs]
    probs_ref, cache = O.forward(spec, p64, x64, mask, theta, want_cache=True)
    probs = m.predict(inputs, mask, theta)
    for s in range(3):
        for l in range(4):
            assert np.abs(m.encoder_activation(s, l, B, T) - cache["streams"][s]["acts"][l + 1]).max() <= 1e-4, (s, l)
    err = np.abs(probs - probs_ref).max()
    print("bf16x3 vs fp64 oracle, real widths: max |dp| = %.2e" % err)
    assert err <= 1e-4
    np.testing.assert_array_equal(O.majority_vote(probs, mask), O.majority_vote(probs_ref, mask))
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, x64, y, mask, theta)
    l = m.compute_grads(inputs, y, mask, theta)
    assert abs(l - l_ref) <= 1e-5 * abs(l_ref)
    g = m.get_grads_dict()
    gscale = max(np.abs(v).max() for v in g_ref.values())
    worst = 0.0
    for k in O.param_names(spec):
        e = np.abs(g[k] - g_ref[k]).max() / max(np.abs(g_ref[k]).max(), 1e-3 * gscale)
        worst = max(worst, e)
        assert e <= 2e-4, (k, e)
    print("bf16x3 gradients: worst tensor error %.2e of its scale" % worst)
    m.close()


def test_bf16x3_at_the_bench_geometry_against_f32_mode(torch_cuda, lib):
    """B = 520, T = 40 (every GEMM large enough to take the split path): probabilities within 1e-4 of the fp32 mode's, identical
    votes; the loss to 1e-5."""
    import bench
    from ip_avsr_amd.model import AdeNetModel
    torch = torch_cuda
    m = AdeNetModel(bench.build_spec())
    bench.synthetic_params(m)
    xs, y, m_d, mask = bench.synthetic_batch(torch, 0, bench.B_PER_GPU, torch.device("cuda", 0))
    for _ in range(2):
        m.train_step(xs, y, m_d, bench.THETA, 2e-3, want_loss=False)
    out = {}
    for prec in ("f32", "bf16x3"):
        m.set_precision(prec)
        out[prec] = (m.predict(xs, m_d, bench.THETA), m.compute_grads(xs, y, m_d, bench.THETA), m.get_grads_dict())
    err = np.abs(out["f32"][0] - out["bf16x3"][0]).max()
    print("bench geometry, bf16x3 vs f32 mode: max |dp| = %.2e" % err)
    assert err <= 1e-4
    np.testing.assert_array_equal(O.majority_vote(out["f32"][0], mask), O.majority_vote(out["bf16x3"][0], mask))
    assert abs(out["f32"][1] - out["bf16x3"][1]) <= 1e-5 * abs(out["f32"][1])
    # Gradients as whole tensors.  Above the rectifiers (bottleneck, LSTMs, classifier) the two modes agree to ~3e-6.  Below
    # them a rectifier input that sits within rounding of zero flips its act' mask in one mode -- ~3e-6 of 10 M units per
    # layer, each changing one element of dZ by 100 %: relative L2 = sqrt(flipped share) ~ 2e-3 (measured 3e-4 ... 3e-3;
    # the fp32 mode against the fp64 oracle shows the same) -- a property of the kink, not of the arithmetic.
    rel = lambda k: (lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(a))(out["f32"][2][k].astype(np.float64).ravel(),
                                                                                out["bf16x3"][2][k].astype(np.float64).ravel())
    for k in ("bottleneck_s1.W", "bottleneck_s3.W", "lstm_s1.W_hid_to_cell", "lstm_s2.W_in_to_ingate", "f_lstm_agg.W_in_to_forgetgate",
              "b_lstm_agg.W_hid_to_outgate", "softmax.W"):
        assert rel(k) <= 5e-5, (k, rel(k))
    for k in ("fc1_s1.W", "fc2_s2.W", "fc3_s3.W", "fc3_s1.b"):
        assert rel(k) <= 1e-2, (k, rel(k))
    m.close()
