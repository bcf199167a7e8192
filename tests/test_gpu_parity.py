"""GPU parity tests proper: the HIP path (called through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerances are written next to each check; the north-star gates are 1e-4 on fp32
encoder activations and identical top-1 per utterance.  Run with ``-m gpu`` on an MI355X."""
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
    assert l.adn_device_count() >= 1, "no gfx950 device visible to libadenet_hip"
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


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(1e-30, np.abs(b).max())


# ============================================================================= operators
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 50, 19), (64, 64, 32), (130, 250, 150), (300, 1000, 250),
                                   (1040, 2000, 1200), (500, 50, 20800), (2600, 130, 66)])
def test_gemm_f32(torch_cuda, lib, layout, M, N, K):
    torch = torch_cuda
    rng = np.random.default_rng(M * 7 + N * 3 + K + layout)
    A = rng.normal(size=(M, K)); Bm = rng.normal(size=(K, N)); bias = rng.normal(size=(N,))
    ref = A @ Bm
    pad = lambda n: (n + 7) // 8 * 8
    if layout == 0:
        a_h, b_h = A, Bm
    elif layout == 1:
        a_h, b_h = A, Bm.T
    else:
        a_h, b_h = A.T, Bm

    def dev(x):
        buf = np.full((x.shape[0], pad(x.shape[1])), np.nan, np.float32)    # poison the padding
        buf[:, :x.shape[1]] = x
        return torch.tensor(buf, device="cuda")

    a_d, b_d = dev(a_h), dev(b_h)
    c_d = torch.full((M, pad(N)), 7.0, device="cuda")
    from ip_avsr_amd._lib import check
    check(lib.adn_op_gemm(layout, M, N, K, dptr(a_d), a_d.shape[1], dptr(b_d), b_d.shape[1], dptr(c_d), c_d.shape[1],
                          None, 0, 0, None))
    torch.cuda.synchronize()
    out = c_d.cpu().numpy()
    tol = 2e-6 * np.sqrt(K) * np.abs(A).max() * np.abs(Bm).max() + 1e-6
    assert np.abs(out[:, :N] - ref).max() <= tol * 4
    assert (out[:, N:] == 7.0).all(), "GEMM wrote outside its N columns"
    # bias + relu epilogue, then accumulate on top
    if layout == 0:
        bias_d = torch.tensor(bias.astype(np.float32), device="cuda")
        check(lib.adn_op_gemm(0, M, N, K, dptr(a_d), a_d.shape[1], dptr(b_d), b_d.shape[1], dptr(c_d), c_d.shape[1],
                              dptr(bias_d), 1, 0, None))
        check(lib.adn_op_gemm(0, M, N, K, dptr(a_d), a_d.shape[1], dptr(b_d), b_d.shape[1], dptr(c_d), c_d.shape[1],
                              None, 0, 1, None))
        torch.cuda.synchronize()
        want = np.maximum(ref + bias, 0) + ref
        assert np.abs(c_d.cpu().numpy()[:, :N] - want).max() <= tol * 8


def test_gemm_identity_with_asymmetric_b_catches_transposes(torch_cuda, lib):
    torch = torch_cuda
    from ip_avsr_amd._lib import check
    n = 96
    eye = torch.eye(n, device="cuda")
    Bm = torch.arange(n * n, device="cuda", dtype=torch.float32).reshape(n, n).contiguous()
    out = torch.zeros(n, n, device="cuda")
    check(lib.adn_op_gemm(0, n, n, n, dptr(eye), n, dptr(Bm), n, dptr(out), n, None, 0, 0, None))
    assert torch.equal(out, Bm)
    check(lib.adn_op_gemm(1, n, n, n, dptr(eye), n, dptr(Bm), n, dptr(out), n, None, 0, 0, None))
    assert torch.equal(out, Bm.T)
    check(lib.adn_op_gemm(2, n, n, n, dptr(Bm), n, dptr(eye), n, dptr(out), n, None, 0, 0, None))
    assert torch.equal(out, Bm.T)


@pytest.mark.parametrize("B,T,F,theta", [(1, 1, 1, 1), (3, 7, 5, 2), (5, 40, 50, 9), (4, 12, 90, 3), (2, 5, 33, 9)])
def test_delta_layer(torch_cuda, lib, B, T, F, theta):
    torch = torch_cuda
    from ip_avsr_amd._lib import check
    rng = np.random.default_rng(B + T + F + theta)
    x = rng.normal(size=(B, T, F)).astype(np.float32)
    ld_in, ld_out = (F + 7) // 8 * 8, (3 * F + 7) // 8 * 8
    xin = np.zeros((B, T, ld_in), np.float32); xin[..., :F] = x
    x_d = torch.tensor(xin, device="cuda")
    out_d = torch.zeros(T, B, ld_out, device="cuda")
    check(lib.adn_op_delta_forward(dptr(x_d), ld_in, dptr(out_d), ld_out, B, T, F, theta, None))
    got = out_d.cpu().numpy()[..., :3 * F].transpose(1, 0, 2)
    want = O.delta_append(x, theta)
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    lit = np.stack([O.append_delta_literal(x[b], theta) for b in range(B)])   # literal scans of the reference
    assert np.abs(got - lit).max() <= 5e-5 * max(1.0, np.abs(lit).max())
    g = rng.normal(size=(B, T, 3 * F)).astype(np.float32)
    gin = np.zeros((T, B, ld_out), np.float32); gin[..., :3 * F] = g.transpose(1, 0, 2)
    g_d = torch.tensor(gin, device="cuda")
    dx_d = torch.zeros(B, T, ld_in, device="cuda")
    check(lib.adn_op_delta_backward(dptr(g_d), ld_out, dptr(dx_d), ld_in, B, T, F, theta, None))
    want_dx = O.delta_append_bwd(g.astype(np.float64), theta)
    assert np.abs(dx_d.cpu().numpy()[..., :F] - want_dx).max() <= 2e-5 * max(1.0, np.abs(want_dx).max())


def test_adam_kernel(torch_cuda, lib):
    torch = torch_cuda
    from ip_avsr_amd._lib import check
    rng = np.random.default_rng(0)
    n = 1003
    p = {"w": rng.normal(size=n).astype(np.float32)}
    st = O.adam_init(p)
    pd = torch.tensor(p["w"], device="cuda"); md = torch.zeros(n, device="cuda"); vd = torch.zeros(n, device="cuda")
    for t in range(1, 4):
        g = rng.normal(size=n).astype(np.float32)
        O.adam_step(p, {"w": g}, st, lr=0.01)
        a_t = np.float32(0.01) * np.sqrt(np.float32(1) - np.float32(0.999) ** np.float32(t)) / (
            np.float32(1) - np.float32(0.9) ** np.float32(t))
        check(lib.adn_op_adam(dptr(pd), dptr(torch.tensor(g, device="cuda")), dptr(md), dptr(vd), n, float(a_t), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(pd.cpu().numpy(), p["w"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(vd.cpu().numpy(), st["v"]["w"], rtol=2e-6, atol=1e-9)


# ============================================================================= whole model
def make_case(spec, B, T, seed, enc_std=0.3, perturb=0.1):
    rng = np.random.default_rng(seed)
    p = O.init_params(spec, rng, np.float32, enc_std=enc_std, perturb=perturb)
    mask = ragged_mask(rng, B, T)
    inputs = [(rng.normal(size=(B, T, s["input_dim"])) * mask[..., None]).astype(np.float32) for s in spec["streams"]]
    y = np.repeat(rng.integers(0, spec["classes"], size=(B, 1)), T, axis=1).astype(np.int32)
    return p, inputs, y, mask


def small_specs():
    out = {}
    out["3stream_concat"] = O.spec_nstream([12, 9, 10], enc_shapes=(14, 6), enc_acts=("rectify", "linear"),
                                           lstm_size=10, classes=5, fusion="concat")
    out["2stream_sum_peep"] = O.spec_nstream([12, 9], enc_shapes=(14, 6), enc_acts=("sigmoid", "linear"),
                                             lstm_size=7, classes=4, fusion="sum", peepholes=True)
    out["3stream_adasum_peep"] = O.spec_nstream([8, 8, 6], enc_shapes=(9, 5), enc_acts=("tanh", "rectify"),
                                                lstm_size=9, classes=3, fusion="adasum", peepholes=True)
    v2 = O.spec_nstream([12, 9], enc_shapes=(14, 6), enc_acts=("rectify", "linear"), lstm_size=8, classes=6,
                        fusion="sum", has_encoder=[True, False])                 # adenet_v2: encoder + raw DCT stream
    out["adenet_v2_like"] = v2
    out["deltanet_blstm"] = O.spec_deltanet(11, enc_shapes=(13, 5), enc_acts=("rectify", "linear"), lstm_size=6,
                                            classes=4, peepholes=True, use_blstm=True)
    out["deltanet_lstm"] = O.spec_deltanet(11, enc_shapes=(13, 5), enc_acts=("leaky_rectify", "linear"), lstm_size=6,
                                           classes=4, use_blstm=False)
    nd = O.spec_deltanet(10, enc_shapes=(), enc_acts=(), lstm_size=5, classes=3, use_blstm=True)
    nd["streams"][0]["delta"] = False                                            # lstm_classifier_majority_vote
    out["lstm_classifier"] = nd
    return out


@pytest.mark.parametrize("name", list(small_specs()))
def test_forward_loss_grads_match_oracle(torch_cuda, lib, name):
    from ip_avsr_amd.model import AdeNetModel
    spec = small_specs()[name]
    B, T, theta = 5, 9, 3
    p, inputs, y, mask = make_case(spec, B, T, seed=sum(map(ord, name)))
    m = AdeNetModel(spec)
    assert [q.name for q in m.params] == O.param_names(spec)            # Lasagne order
    for q in m.params:
        assert q.shape == tuple(O.param_shapes(spec)[q.name]), q.name
    m.set_params_dict(p)
    back = m.get_params_dict()
    for k in p:
        np.testing.assert_array_equal(back[k], p[k])                       # set/get round trip is exact
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    in64 = [x.astype(np.float64) for x in inputs]
    probs_ref = O.forward(spec, p64, in64, mask, theta)
    probs = m.predict(inputs, mask, theta)
    assert probs.shape == (B, T, spec["classes"]) and probs.dtype == np.float32
    assert np.abs(probs - probs_ref).max() <= 2e-5
    loss_ref, g_ref, cache = O.loss_and_grads(spec, p64, in64, y, mask, theta)
    assert abs(m.loss(inputs, y, mask, theta) - loss_ref) <= 1e-5 * abs(loss_ref)
    loss = m.compute_grads(inputs, y, mask, theta)
    assert abs(loss - loss_ref) <= 1e-5 * abs(loss_ref)
    grads = m.get_grads_dict()
    gscale = max(np.abs(v).max() for v in g_ref.values())
    for k in O.param_names(spec):
        err = np.abs(grads[k] - g_ref[k]).max()
        assert err <= 1e-4 * max(np.abs(g_ref[k]).max(), 1e-3 * gscale) + 1e-9, (k, err, np.abs(g_ref[k]).max())
    # encoder activations (north-star gate: 1e-4)
    for s, (ss, sc) in enumerate(zip(spec["streams"], cache["streams"])):
        for l in range(len(ss["enc_shapes"])):
            assert np.abs(m.encoder_activation(s, l, B, T) - sc["acts"][l + 1]).max() <= 1e-4
    m.close()


@pytest.mark.parametrize("name", ["3stream_concat", "3stream_adasum_peep", "deltanet_blstm"])
def test_train_steps_match_oracle(torch_cuda, lib, name):
    from ip_avsr_amd.model import AdeNetModel
    spec = small_specs()[name]
    B, T, theta, lr = 6, 8, 2, 1e-3
    p, inputs, y, mask = make_case(spec, B, T, seed=11)
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    st = O.adam_init(p64)
    in64 = [x.astype(np.float64) for x in inputs]
    for step in range(4):
        l_ref = O.train_step(spec, p64, st, in64, y, mask, theta, lr)
        l = m.train_step(inputs, y, mask, theta, lr)
        assert abs(l - l_ref) <= 2e-5 * abs(l_ref), (step, l, l_ref)
    got = m.get_params_dict()
    for k in p64:
        # Adam's first steps move every weight by ~lr regardless of gradient scale, so compare against lr
        assert np.abs(got[k] - p64[k]).max() <= 0.05 * lr, k
    state = m.get_adam_state()
    assert state["t"] == 4
    m.close()


def test_grad_clip_fires_like_the_oracle(torch_cuda, lib):
    """Large weights -> saturating BPTT -> the +-5 clip on d(gates) must bind identically."""
    from ip_avsr_amd.model import AdeNetModel
    spec = O.spec_deltanet(6, enc_shapes=(), enc_acts=(), lstm_size=5, classes=3, use_blstm=False)
    B, T, theta = 4, 7, 1
    p, inputs, y, mask = make_case(spec, B, T, seed=3)
    rng = np.random.default_rng(0)
    p["softmax.W"] = (rng.normal(size=p["softmax.W"].shape) * 4000).astype(np.float32)
    inputs = [x * 1 for x in inputs]
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    _, g_ref, cache = O.loss_and_grads(spec, p64, [x.astype(np.float64) for x in inputs], y, mask, theta)
    m.compute_grads(inputs, y, mask, theta)
    g = m.get_grads_dict()
    for k in ("lstm.b_ingate", "lstm.b_outgate", "lstm.W_hid_to_cell"):
        assert np.abs(g[k] - g_ref[k]).max() <= 2e-4 * max(1.0, np.abs(g_ref[k]).max()), k
    m.close()


def test_real_dimensions_encoder_1e4_and_top1(torch_cuda, lib):
    """AVLetters-shaped 3-stream concat model (1200-2000-1000-500-50, H=250, C=26, theta=9)."""
    from ip_avsr_amd.model import AdeNetModel
    spec = O.spec_nstream([1200, 1200, 1200])
    B, T, theta = 10, 14, 9
    rng = np.random.default_rng(1234)
    p = O.init_params(spec, rng, np.float32, enc_std=0.01)
    for k in p:                                          # non-degenerate biases
        if k.endswith(".b"):
            p[k] = rng.normal(0, 0.05, p[k].shape).astype(np.float32)
    mask = ragged_mask(rng, B, T)
    inputs = [(rng.normal(size=(B, T, 1200)) * mask[..., None]).astype(np.float32) for _ in range(3)]
    y = np.repeat((np.arange(B) % 26)[:, None], T, axis=1).astype(np.int32)
    m = AdeNetModel(spec)
    assert m.count_params() == 17999676
    m.set_params_dict(p)
    probs_ref, cache = O.forward(spec, p, inputs, mask, theta, want_cache=True)   # fp32 oracle, like the reference
    probs = m.predict(inputs, mask, theta)
    for s in range(3):
        for l in range(4):
            ref = cache["streams"][s]["acts"][l + 1]
            assert np.abs(m.encoder_activation(s, l, B, T) - ref).max() <= 1e-4, (s, l)
    assert np.abs(probs - probs_ref).max() <= 1e-4
    np.testing.assert_array_equal(O.majority_vote(probs, mask), O.majority_vote(probs_ref, mask))
    # one training step at full width
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, [x.astype(np.float64) for x in inputs], y, mask, theta)
    l = m.compute_grads(inputs, y, mask, theta)
    assert abs(l - l_ref) <= 1e-5 * abs(l_ref)
    g = m.get_grads_dict()
    gscale = max(np.abs(v).max() for v in g_ref.values())
    for k in O.param_names(spec):
        err = np.abs(g[k] - g_ref[k]).max()
        assert err <= 2e-4 * max(np.abs(g_ref[k]).max(), 1e-3 * gscale), (k, err)
    m.close()


def test_device_inputs_equal_host_inputs(torch_cuda, lib):
    torch = torch_cuda
    from ip_avsr_amd.model import AdeNetModel
    spec = small_specs()["3stream_concat"]
    p, inputs, y, mask = make_case(spec, 4, 6, seed=5)
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    host = m.predict(inputs, mask, 2)
    dev = m.predict([torch.tensor(x, device="cuda") for x in inputs], torch.tensor(mask, device="cuda"), 2)
    np.testing.assert_array_equal(host, dev)
    l_host = m.compute_grads(inputs, y, mask, 2)
    l_dev = m.compute_grads([torch.tensor(x, device="cuda") for x in inputs], torch.tensor(y, device="cuda"),
                            torch.tensor(mask, device="cuda"), 2)
    assert l_host == l_dev
    m.close()


def test_edge_cases_single_frame_single_utterance_and_errors(torch_cuda, lib):
    from ip_avsr_amd.model import AdeNetModel, AdenetError
    spec = small_specs()["3stream_concat"]
    p, inputs, y, mask = make_case(spec, 1, 1, seed=8)
    mask[:] = 1
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    ref = O.forward(spec, p64, [x.astype(np.float64) for x in inputs], mask, 9)
    assert np.abs(m.predict(inputs, mask, 9) - ref).max() <= 2e-5        # T=1 with theta=9: all-clamped deltas
    with pytest.raises(ValueError):
        m.predict(inputs[:2], mask, 3)                                    # wrong stream count
    with pytest.raises(ValueError):
        m.predict([x[:, :, :-1] for x in inputs], mask, 3)                # wrong feature width
    with pytest.raises(AdenetError):
        m.apply_adam(0.1)                                                 # no gradients yet
    with pytest.raises(ValueError):
        m.set_all_param_values(m.get_all_param_values()[:-1])
    m.close()


# ============================================================================= full-size properties
def test_full_size_properties_whole_train_batch(torch_cuda, lib):
    """BASELINE config 2 size (B=520, T=40): properties that need no oracle run --
    (i) utterance permutation equivariance, (ii) gradient additivity over utterance shards with a
    common normaliser (the data-parallel identity), (iii) probabilities sum to 1."""
    torch = torch_cuda
    from ip_avsr_amd.model import AdeNetModel
    spec = O.spec_nstream([1200, 1200, 1200])
    B, T, theta = 520, 40, 9
    rng = np.random.default_rng(7)
    p = O.init_params(spec, rng, np.float32, enc_std=0.01)
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    lens = rng.integers(12, 41, size=B); lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    gen = torch.Generator(device="cuda").manual_seed(1)
    xs = [torch.randn(B, T, 1200, device="cuda", generator=gen) * torch.tensor(mask, device="cuda")[..., None]
          for _ in range(3)]
    y = np.repeat((np.arange(B) % 26)[:, None], T, axis=1).astype(np.int32)
    probs = m.predict(xs, mask, theta)
    assert np.isfinite(probs).all() and np.abs(probs.sum(-1) - 1).max() < 1e-5
    perm = rng.permutation(B)
    pt = torch.tensor(perm, device="cuda")
    probs_p = m.predict([x[pt].contiguous() for x in xs], mask[perm], theta)
    assert np.abs(probs_p - probs[perm]).max() <= 1e-6
    total = float(mask.sum())
    l_full = m.compute_grads(xs, y, mask, theta)
    g_full = m.get_grads_dict()
    half = B // 2
    parts, losses = [], []
    for sl in (slice(0, half), slice(half, B)):
        losses.append(m.compute_grads([x[sl].contiguous() for x in xs], y[sl], mask[sl], theta, total_frames=total))
        parts.append(m.get_grads_dict())
    assert abs(losses[0] + losses[1] - l_full) <= 1e-5 * abs(l_full)
    for k in g_full:
        s = parts[0][k] + parts[1][k]
        assert np.abs(s - g_full[k]).max() <= 1e-4 * max(np.abs(g_full[k]).max(), 1e-7), k
    m.close()


# ============================================================================= bf16 GEMM mode
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(37, 50, 19), (130, 250, 150), (1040, 2000, 1200), (500, 50, 20800)])
def test_gemm_bf16_matches_bf16_rounded_reference(torch_cuda, lib, layout, M, N, K):
    """The bf16 kernel must equal an fp32-accumulated product of the bf16-ROUNDED operands (tight), which
    separates rounding-of-inputs (expected) from layout / indexing mistakes (bugs)."""
    torch = torch_cuda
    from ip_avsr_amd import _lib as L
    rng = np.random.default_rng(M + N + K + layout)
    A = rng.normal(size=(M, K)).astype(np.float32); Bm = rng.normal(size=(K, N)).astype(np.float32)
    a_d, b_d = torch.tensor(A, device="cuda"), torch.tensor(Bm, device="cuda")
    ref = (a_d.bfloat16().double() @ b_d.bfloat16().double()).cpu().numpy()
    pad = lambda n: (n + 7) // 8 * 8

    def dev(x):
        buf = torch.full((x.shape[0], pad(x.shape[1])), float("nan"), device="cuda")
        buf[:, :x.shape[1]] = x
        return buf
    if layout == 0:
        ah, bh = a_d, b_d
    elif layout == 1:
        ah, bh = a_d, b_d.T.contiguous()
    else:
        ah, bh = a_d.T.contiguous(), b_d
    ap, bp = dev(ah), dev(bh)
    c_d = torch.full((M, pad(N)), 7.0, device="cuda")
    L.check(lib.adn_op_gemm_ex(layout, M, N, K, dptr(ap), ap.shape[1], dptr(bp), bp.shape[1], dptr(c_d), c_d.shape[1],
                               None, 0, 0, L.PRECISION["bf16"], None))
    torch.cuda.synchronize()
    out = c_d.cpu().numpy()
    assert np.abs(out[:, :N] - ref).max() <= 1e-5 * np.sqrt(K) * 16 + 1e-5
    assert (out[:, N:] == 7.0).all()


def test_bf16_mode_tracks_f32_mode_and_keeps_top1(torch_cuda, lib):
    """bf16 GEMM arithmetic: same parameters, same inputs; probabilities close, majority-vote decisions
    identical on separable data, gradients aligned (cosine) with the fp32 ones."""
    from ip_avsr_amd.model import AdeNetModel
    spec = O.spec_nstream([1200, 1200, 1200])
    B, T, theta = 26, 20, 9
    rng = np.random.default_rng(99)
    p = O.init_params(spec, rng, np.float32, enc_std=0.01)
    mask = ragged_mask(rng, B, T)
    inputs = [(rng.normal(size=(B, T, 1200)) * mask[..., None]).astype(np.float32) for _ in range(3)]
    y = np.repeat((np.arange(B) % 26)[:, None], T, axis=1).astype(np.int32)
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    for _ in range(30):                                   # a few fp32 steps so that the outputs are not flat
        m.train_step(inputs, y, mask, theta, 2e-3)
    probs32 = m.predict(inputs, mask, theta)
    l32 = m.compute_grads(inputs, y, mask, theta)
    g32 = m.get_grads_dict()
    m.set_precision("bf16")
    probs16 = m.predict(inputs, mask, theta)
    l16 = m.compute_grads(inputs, y, mask, theta)
    g16 = m.get_grads_dict()
    assert np.abs(probs16 - probs32).max() < 0.05
    assert abs(l16 - l32) < 2e-2 * abs(l32)
    v32, v16 = O.majority_vote(probs32, mask), O.majority_vote(probs16, mask)
    assert (v32 == v16).mean() >= 0.95
    for k in ("fc1_s1.W", "bottleneck_s2.W", "lstm_s3.W_in_to_cell", "f_lstm_agg.W_hid_to_ingate", "softmax.W"):
        a, b = g32[k].ravel().astype(np.float64), g16[k].ravel().astype(np.float64)
        cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
        assert cos > 0.98, (k, cos)
    m.close()


def test_bf16_shadow_copies_equal_convert_in_flight(torch_cuda, lib):
    """bf16 mode keeps bf16 shadow copies of every GEMM operand; rounding the same fp32 value gives the same
    bf16, so the result must be IDENTICAL to the path that converts in flight (ADN_BF16_NO_SHADOW=1).  A stale
    or missing shadow refresh anywhere in the graph shows up here as a difference."""
    import os
    from ip_avsr_amd.model import AdeNetModel
    for name in ("3stream_concat", "3stream_adasum_peep", "deltanet_blstm", "adenet_v2_like", "2stream_sum_peep"):
        spec = dict(small_specs()[name], precision="bf16")
        p, inputs, y, mask = make_case(spec, 6, 9, seed=21)
        out = {}
        for mode in ("shadow", "inflight"):
            if mode == "inflight":
                os.environ["ADN_BF16_NO_SHADOW"] = "1"
            else:
                os.environ.pop("ADN_BF16_NO_SHADOW", None)
            try:
                m = AdeNetModel(spec)
                m.set_params_dict(p)
                losses = [m.train_step(inputs, y, mask, 2, 1e-3) for _ in range(3)]
                probs = m.predict(inputs, mask, 2)
                m.compute_grads(inputs, y, mask, 2)
                out[mode] = (losses, probs, m.get_grads_dict())
                m.close()
            finally:
                os.environ.pop("ADN_BF16_NO_SHADOW", None)
        # act'(Y) is evaluated on the bf16 copy of Y in shadow mode: exact for piecewise-linear encoders (the sign
        # of y survives rounding), a bf16-sized difference for sigmoid / tanh ones
        acts = {a for st in spec["streams"] for a in st["enc_acts"]}
        exact = acts <= {"rectify", "linear", "leaky_rectify", "very_leaky_rectify"}
        if exact:
            assert out["shadow"][0] == out["inflight"][0], name
            np.testing.assert_array_equal(out["shadow"][1], out["inflight"][1])
        else:
            np.testing.assert_allclose(out["shadow"][0], out["inflight"][0], rtol=2e-3)
            assert np.abs(out["shadow"][1] - out["inflight"][1]).max() < 5e-3
        for k in out["shadow"][2]:
            a, b = out["shadow"][2][k], out["inflight"][2][k]
            # split-K weight gradients use fp32 atomics (order-dependent last bits)
            tol = 1e-5 if exact else 3e-2
            assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-6), (name, k)


def wide_spec():
    s = O.spec_deltanet(10, enc_shapes=(), enc_acts=(), lstm_size=260, classes=5, peepholes=True, use_blstm=True)
    return s                                                  # H > 256: the 64-units-per-wave persistent kernels


@pytest.mark.parametrize("name", ["lstm_classifier", "deltanet_blstm", "3stream_adasum_peep", "wide"])
def test_bf16_recurrence_close_to_oracle(torch_cuda, lib, name):
    """bf16 mode also runs the recurrent products h*W_hid and dG*W_hid^T on bf16 MFMA (fp32 accumulate, fp32
    cell state / gate math).  Against the fp64 oracle: probabilities within 2e-2, every gradient tensor
    aligned (cosine > 0.99) and of the right size (norm ratio within 15 %)."""
    from ip_avsr_amd.model import AdeNetModel
    spec = dict(wide_spec() if name == "wide" else small_specs()[name], precision="bf16")
    B, T, theta = (19, 11, 2) if name == "wide" else (7, 11, 2)
    p, inputs, y, mask = make_case(spec, B, T, seed=77)
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    in64 = [x.astype(np.float64) for x in inputs]
    probs_ref = O.forward(spec, p64, in64, mask, theta)
    assert np.abs(m.predict(inputs, mask, theta) - probs_ref).max() <= 2e-2
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, in64, y, mask, theta)
    l = m.compute_grads(inputs, y, mask, theta)
    assert abs(l - l_ref) <= 1e-2 * abs(l_ref)
    g = m.get_grads_dict()
    for k in O.param_names(spec):
        a, b = np.asarray(g_ref[k], np.float64).ravel(), g[k].ravel().astype(np.float64)
        if np.linalg.norm(a) < 1e-9:
            continue
        cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
        assert cos > 0.99 and abs(np.linalg.norm(b) / np.linalg.norm(a) - 1) < 0.15, (k, cos)
    m.close()


def test_adam_with_per_layer_learning_rates(torch_cuda, lib):
    """custom/updates.py generate_lr_map + adam_vlr: per-layer rates, one step counter."""
    from ip_avsr_amd.custom.updates import adam_vlr, generate_lr_map
    from ip_avsr_amd.model import AdeNetModel, AdenetError
    spec = small_specs()["3stream_adasum_peep"]
    p, inputs, y, mask = make_case(spec, 5, 7, seed=31)
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    lr_config = {"fc1_s1": 1e-4, "bottleneck_s2": 5e-3, "lstm_s3": 2e-3, "softmax": 0.0}
    lr_map = generate_lr_map(m.get_all_params(trainable=True), lr_config, 1e-3)
    step = adam_vlr(m, lr_map)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    st = O.adam_init(p64)
    in64 = [x.astype(np.float64) for x in inputs]
    for it in range(3):
        _, g, _ = O.loss_and_grads(spec, p64, in64, y, mask, 2)
        st["t"] += 1
        t = st["t"]
        for k in p64:                                           # oracle: the same formula with lr looked up per layer
            lr = lr_config.get(k[:k.rfind(".")], 1e-3)
            a_t = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
            st["m"][k] = 0.9 * st["m"][k] + 0.1 * g[k]
            st["v"][k] = 0.999 * st["v"][k] + 0.001 * g[k] * g[k]
            p64[k] = p64[k] - a_t * st["m"][k] / (np.sqrt(st["v"][k]) + 1e-8)
        m.compute_grads(inputs, y, mask, 2)
        step()
    got = m.get_params_dict()
    for k in p64:
        lr = lr_config.get(k[:k.rfind(".")], 1e-3)
        assert np.abs(got[k] - p64[k]).max() <= 0.05 * max(lr, 1e-7) + 1e-7, k
    np.testing.assert_array_equal(got["softmax.W"], p["softmax.W"])            # rate 0: untouched
    # one LSTM's gates live in one tensor: they cannot get different rates
    bad = dict(lr_map)
    bad[[q for q in m.params if q.name == "lstm_s1.W_in_to_cell"][0]] = 0.5
    m.compute_grads(inputs, y, mask, 2)
    with pytest.raises(AdenetError):
        m.apply_adam_vlr(bad)
    m.close()


def test_four_streams_512_units_config5_shape(torch_cuda, lib):
    """BASELINE configs[4] topology: adenet_4stream, concat fusion, 512-unit LSTMs (small encoders / batch so
    that the fp64 oracle stays fast).  f32 and bf16x3 modes against the oracle; bf16 mode must track it."""
    from ip_avsr_amd.model import AdeNetModel
    spec = O.spec_nstream([40, 36, 44, 30], enc_shapes=(48, 24, 10), enc_acts=("rectify", "rectify", "linear"),
                          lstm_size=512, classes=10, fusion="concat")
    B, T, theta = 6, 8, 3
    p, inputs, y, mask = make_case(spec, B, T, seed=123, enc_std=0.2, perturb=0.02)
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    in64 = [x.astype(np.float64) for x in inputs]
    probs_ref = O.forward(spec, p64, in64, mask, theta)
    assert np.abs(m.predict(inputs, mask, theta) - probs_ref).max() <= 5e-5
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, in64, y, mask, theta)
    l = m.compute_grads(inputs, y, mask, theta)
    assert abs(l - l_ref) <= 1e-5 * abs(l_ref)
    g = m.get_grads_dict()
    gscale = max(np.abs(v).max() for v in g_ref.values())
    for k in O.param_names(spec):
        assert np.abs(g[k] - g_ref[k]).max() <= 2e-4 * max(np.abs(g_ref[k]).max(), 1e-3 * gscale), k
    m.set_precision("bf16")
    assert np.abs(m.predict(inputs, mask, theta) - probs_ref).max() <= 3e-2
    l16 = m.compute_grads(inputs, y, mask, theta)
    assert abs(l16 - l_ref) <= 2e-2 * abs(l_ref)
    # the parity-grade mode on this topology: the 16-workgroup bf16x3 LSTM kernels (csrc/lstm_cluster.hip, *_x3w_*) under the
    # fp32 mode's gate
    m.set_precision("bf16x3")
    assert np.abs(m.predict(inputs, mask, theta) - probs_ref).max() <= 5e-5
    l3 = m.compute_grads(inputs, y, mask, theta)
    assert abs(l3 - l_ref) <= 1e-5 * abs(l_ref)
    g = m.get_grads_dict()
    for k in O.param_names(spec):
        assert np.abs(g[k] - g_ref[k]).max() <= 2e-4 * max(np.abs(g_ref[k]).max(), 1e-3 * gscale), k
    m.close()


@pytest.mark.parametrize("D,C,B", [(1500, 10, 10), (1144, 10, 10)])
def test_cuave_oulu_input_widths(torch_cuda, lib, D, C, B):
    """BASELINE configs[2]/[3] input geometry: CUAVE 30x50 = 1500 (not a multiple of 8), OuluVS 26x44 = 1144;
    2-stream sum fusion, 10 classes, batch 10."""
    from ip_avsr_amd.model import AdeNetModel
    spec = O.spec_nstream([D, 90], fusion="sum", classes=C, has_encoder=[True, False])       # adenet_v2 topology
    T, theta = 9, 9
    rng = np.random.default_rng(D)
    p = O.init_params(spec, rng, np.float32, enc_std=0.01)
    for k in p:
        if k.endswith(".b"):
            p[k] = rng.normal(0, 0.05, p[k].shape).astype(np.float32)
    mask = ragged_mask(rng, B, T)
    inputs = [(rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32) for d in (D, 90)]
    y = np.repeat((np.arange(B) % C)[:, None], T, axis=1).astype(np.int32)
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    probs_ref, cache = O.forward(spec, p, inputs, mask, theta, want_cache=True)
    probs = m.predict(inputs, mask, theta)
    for l in range(4):
        assert np.abs(m.encoder_activation(0, l, B, T) - cache["streams"][0]["acts"][l + 1]).max() <= 1e-4
    assert np.abs(probs - probs_ref).max() <= 1e-4
    np.testing.assert_array_equal(O.majority_vote(probs, mask), O.majority_vote(probs_ref, mask))
    for prec in ("f32", "bf16"):                                # both staging paths (D % 8 != 0 -> padded copy)
        m.set_precision(prec)
        import torch
        pd = m.predict([torch.tensor(x, device="cuda") for x in inputs], torch.tensor(mask, device="cuda"), theta)
        assert np.abs(pd - probs_ref).max() <= (1e-4 if prec == "f32" else 3e-2)
    m.close()


@pytest.mark.parametrize("H", [37, 300, 500])
def test_weight_stationary_lstm_equals_single_workgroup_path(torch_cuda, lib, monkeypatch, H):
    """bf16 mode: the LSTMs run on groups of 4 (H <= 256) or 8 (H <= 512: adenet_v3's 500 units, the 4-stream model's
    512) workgroups that keep W_hid in LDS / registers and exchange h / partial dh
    through tagged granules (csrc/lstm_cluster.hip).  Same arithmetic as the one-workgroup-per-slice kernels
    (csrc/lstm_persistent.hip, selected with ADN_LSTM_NO_CLUSTER): forward identical, gradients equal up to the 19-bit
    partial sums of the backward exchange.  B = 70 spans three 32-utterance groups with a ragged last one; the launches
    are repeated so that stale granules of earlier launches (tags, emptied inboxes) would be noticed; and the
    bidirectional + peephole LSTMs exercise both directions.  The oracle check bounds both paths.  At H > 256 the
    one-workgroup kernel is no longer a production path (ADN_LSTM_WIDE_PERSISTENT selects it); what runs when a launch
    cannot keep a whole LSTM resident are the per-step kernels (lstm.hip), compared here as well."""
    from ip_avsr_amd.model import AdeNetModel
    spec = dict(O.spec_nstream([12, 9], enc_shapes=(14, 6), enc_acts=("rectify", "linear"), lstm_size=H, classes=5,
                               fusion="concat", peepholes=True), precision="bf16")
    B, T, theta = 70, 13, 2
    p, inputs, y, mask = make_case(spec, B, T, seed=4242, perturb=0.1 if H <= 256 else 0.02)   # (keeps the gates unsaturated)
    results = {}
    for mode in ("cluster", "single", "steps"):
        monkeypatch.delenv("ADN_LSTM_NO_CLUSTER", raising=False)
        monkeypatch.delenv("ADN_LSTM_WIDE_PERSISTENT", raising=False)
        if mode != "cluster":
            monkeypatch.setenv("ADN_LSTM_NO_CLUSTER", "1")
        if mode == "single":
            monkeypatch.setenv("ADN_LSTM_WIDE_PERSISTENT", "1")
        if mode == "steps" and H <= 256:
            continue
        m = AdeNetModel(spec)
        m.set_params_dict(p)
        runs = []
        for rep in range(3):                                   # same inputs again: every launch must reproduce itself
            probs = m.predict(inputs, mask, theta)
            loss = m.compute_grads(inputs, y, mask, theta)
            runs.append((probs, loss, m.get_grads_dict()))
        for probs, loss, g in runs[1:]:
            np.testing.assert_array_equal(probs, runs[0][0])
            assert loss == runs[0][1]
            for k in g:
                np.testing.assert_allclose(g[k], runs[0][2][k], rtol=0,
                                           atol=(1e-6 if H <= 256 else 1e-5) * max(1.0, np.abs(runs[0][2][k]).max()))
        results[mode] = runs[0]
        m.close()
    monkeypatch.delenv("ADN_LSTM_NO_CLUSTER", raising=False)
    monkeypatch.delenv("ADN_LSTM_WIDE_PERSISTENT", raising=False)
    pc, lc, gc = results["cluster"]
    ps, ls, gs = results["single"]
    valid = mask[..., None].astype(bool)
    np.testing.assert_array_equal(pc * valid, ps * valid)        # forward: the same products in the same order
    assert abs(lc - ls) <= 1e-6 * abs(ls)
    for other in [results["single"]] + ([results["steps"]] if H > 256 else []):
        for k in gc:
            scale = max(np.abs(other[2][k]).max(), 1e-6)
            assert np.abs(gc[k] - other[2][k]).max() <= 5e-3 * scale, (k, np.abs(gc[k] - other[2][k]).max(), scale)
    if H > 256:
        assert np.abs((pc - results["steps"][0]) * valid).max() <= 1e-6
    probs_ref = O.forward(spec, {k: v.astype(np.float64) for k, v in p.items()}, [x.astype(np.float64) for x in inputs],
                          mask, theta)
    assert np.abs(pc - probs_ref).max() <= 2e-2


def test_concat_as_one_gemm_equals_blockwise_products(torch_cuda, lib, monkeypatch):
    """bf16 mode, concat fusion: the aggregation BLSTM reads one materialised [N][S*ldh] bf16 matrix (input projection,
    dW_in and d(concat) are one GEMM each per LSTM) instead of S column blocks (ADN_NO_CAT=1).  Same products, other
    summation order: probabilities and every gradient agree to bf16-accumulation noise."""
    from ip_avsr_amd.model import AdeNetModel
    spec = dict(small_specs()["3stream_concat"], precision="bf16")
    B, T, theta = 9, 10, 2
    p, inputs, y, mask = make_case(spec, B, T, seed=99)
    out = {}
    for mode in ("cat", "blocks"):
        if mode == "blocks":
            monkeypatch.setenv("ADN_NO_CAT", "1")
        else:
            monkeypatch.delenv("ADN_NO_CAT", raising=False)
        m = AdeNetModel(spec)
        m.set_params_dict(p)
        probs = m.predict(inputs, mask, theta)
        loss = m.compute_grads(inputs, y, mask, theta)
        out[mode] = (probs, loss, m.get_grads_dict())
        m.close()
    monkeypatch.delenv("ADN_NO_CAT", raising=False)
    assert np.abs(out["cat"][0] - out["blocks"][0]).max() <= 2e-3
    assert abs(out["cat"][1] - out["blocks"][1]) <= 1e-3 * abs(out["blocks"][1])
    for k in out["cat"][2]:
        a, b = out["cat"][2][k].ravel().astype(np.float64), out["blocks"][2][k].ravel().astype(np.float64)
        if np.linalg.norm(b) < 1e-9:
            continue
        cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
        assert cos > 0.999 and abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) < 0.02, (k, cos)


def test_bf16_parameter_shadow_written_by_adam_equals_a_fresh_conversion(torch_cuda, lib):
    """bf16 mode: the Adam kernel writes the bf16 copy of the parameters that the next step's GEMMs read (and the
    derived images -- transposed weights, LSTM fragments, concatenated W_in -- are rebuilt from the fp32 values).  A
    model whose parameters are written back through the API after every update (which forces the separate conversion)
    must follow the same trajectory."""
    from ip_avsr_amd.model import AdeNetModel
    spec = dict(small_specs()["3stream_concat"], precision="bf16")
    B, T, theta = 9, 10, 2
    p, inputs, y, mask = make_case(spec, B, T, seed=321)
    a, b = AdeNetModel(spec), AdeNetModel(spec)
    a.set_params_dict(p); b.set_params_dict(p)
    la, lb = [], []
    for step in range(4):
        la.append(float(a.train_step(inputs, y, mask, theta, 1e-2)))
        lb.append(float(b.train_step(inputs, y, mask, theta, 1e-2)))
        b.set_params_dict(b.get_params_dict())                 # marks every derived copy stale, the shadow included
    assert la[-1] < la[0]                                       # it trains
    np.testing.assert_allclose(la, lb, rtol=2e-4)
    pa, pb = a.get_params_dict(), b.get_params_dict()
    for k in pa:
        assert np.abs(pa[k] - pb[k]).max() <= 2e-3 * max(np.abs(pb[k]).max(), 1e-3), k
    a.close(); b.close()


@pytest.mark.parametrize("B,T", [(1, 1), (1, 7), (33, 1), (33, 2), (64, 5)])
def test_weight_stationary_lstm_edge_shapes(torch_cuda, lib, monkeypatch, B, T):
    """bf16 mode, the resident-weight LSTM kernels at the edges: one utterance, one time step (the forward loop never
    polls, the backward epilogue does), 33 utterances = two groups with one row in the second, utterances of one valid
    frame.  Forward and gradients against the one-workgroup kernels (same arithmetic) and the oracle's probabilities."""
    from ip_avsr_amd.model import AdeNetModel
    spec = dict(small_specs()["2stream_sum_peep"], precision="bf16")
    p, inputs, y, mask = make_case(spec, B, T, seed=100 * B + T)
    mask[:, 1:] = mask[:, 1:] * (np.arange(B)[:, None] % 3 != 0)        # every third utterance has a single valid frame
    inputs = [x * mask[..., None] for x in inputs]
    out = {}
    for mode in ("cluster", "single"):
        if mode == "single":
            monkeypatch.setenv("ADN_LSTM_NO_CLUSTER", "1")
        else:
            monkeypatch.delenv("ADN_LSTM_NO_CLUSTER", raising=False)
        m = AdeNetModel(spec)
        m.set_params_dict(p)
        probs = m.predict(inputs, mask, 3)
        loss = m.compute_grads(inputs, y, mask, 3)
        out[mode] = (probs, loss, m.get_grads_dict())
        m.close()
    monkeypatch.delenv("ADN_LSTM_NO_CLUSTER", raising=False)
    valid = mask[..., None].astype(bool)
    np.testing.assert_array_equal(out["cluster"][0] * valid, out["single"][0] * valid)
    assert abs(out["cluster"][1] - out["single"][1]) <= 1e-6 * abs(out["single"][1])
    for k, g in out["cluster"][2].items():
        ref = out["single"][2][k]
        assert np.isfinite(g).all(), k
        assert np.abs(g - ref).max() <= 5e-3 * max(np.abs(ref).max(), 1e-6), k
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    probs_ref = O.forward(spec, p64, [x.astype(np.float64) for x in inputs], mask, 3)
    assert np.abs((out["cluster"][0] - probs_ref) * valid).max() <= 3e-2


@pytest.mark.parametrize("feat,B,T,bidir", [(6, 70, 13, True), (50, 70, 9, False), (50, 33, 1, True), (30, 9, 2, False), (50, 200, 5, True)])
def test_folded_input_projection_equals_the_projection_gemm(torch_cuda, lib, monkeypatch, feat, B, T, bidir):
    """bf16 mode, H <= 256: the weight-stationary forward kernel multiplies x_t W_in + b itself (csrc/lstm_cluster.hip, KXS = 3 for
    up to 96 input features, 5 for up to 160 -- the 3 x 50 delta features of an encoder stream) instead of reading a projection a
    GEMM wrote.  Same bf16 operands, fp32 accumulation over the k-steps in order, bias last: identical probabilities, the same
    gradients (the backward pass is untouched), in both directions of a BLSTM stream, on ragged masks, at T = 1 / 2 (pipeline
    fill only) and across several 32-utterance groups; both against the oracle."""
    from ip_avsr_amd.model import AdeNetModel
    spec = dict(O.spec_nstream([20, 14], enc_shapes=(24, feat), enc_acts=("rectify", "linear"), lstm_size=40, classes=5,
                               fusion="concat", peepholes=True), precision="bf16")
    if bidir:                                        # summed forward / backward pair per stream (use_blstm_substream)
        for k, st in enumerate(spec["streams"]):
            st["lstm_names"] = ["f_lstm_s%d" % (k + 1), "b_lstm_s%d" % (k + 1)]
    p, inputs, y, mask = make_case(spec, B, T, seed=7 * feat + B)
    out = {}
    for mode in ("fold", "gemm"):
        monkeypatch.setenv("ADN_LSTM_FOLD_MIN_B", "1")
        if mode == "gemm":
            monkeypatch.setenv("ADN_LSTM_NO_FOLD", "1")
        else:
            monkeypatch.delenv("ADN_LSTM_NO_FOLD", raising=False)
        m = AdeNetModel(spec)
        m.set_params_dict(p)
        probs = m.predict(inputs, mask, 2)
        probs2 = m.predict(inputs, mask, 2)
        loss = m.compute_grads(inputs, y, mask, 2)
        out[mode] = (probs, loss, m.get_grads_dict())
        np.testing.assert_array_equal(probs, probs2)
        m.close()
    monkeypatch.delenv("ADN_LSTM_NO_FOLD", raising=False)
    monkeypatch.delenv("ADN_LSTM_FOLD_MIN_B", raising=False)
    valid = mask[..., None].astype(bool)
    assert np.abs((out["fold"][0] - out["gemm"][0]) * valid).max() <= 1e-6
    assert abs(out["fold"][1] - out["gemm"][1]) <= 1e-6 * abs(out["gemm"][1])
    for k, g in out["fold"][2].items():
        ref = out["gemm"][2][k]
        assert np.abs(g - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-6), k
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    probs_ref = O.forward(spec, p64, [x.astype(np.float64) for x in inputs], mask, 2)
    assert np.abs((out["fold"][0] - probs_ref) * valid).max() <= 3e-2


@pytest.mark.parametrize("B,T", [(1, 1), (33, 2)])
def test_wide_weight_stationary_lstm_edge_shapes(torch_cuda, lib, monkeypatch, B, T):
    """The 8-workgroup kernels (256 < H <= 512) at the same edges, against the one-workgroup kernel."""
    from ip_avsr_amd.model import AdeNetModel
    spec = dict(O.spec_nstream([12, 9], enc_shapes=(14, 6), enc_acts=("rectify", "linear"), lstm_size=260, classes=5,
                               fusion="sum", peepholes=True), precision="bf16")
    p, inputs, y, mask = make_case(spec, B, T, seed=7 * B + T, perturb=0.02)
    out = {}
    for mode in ("cluster", "single"):
        monkeypatch.delenv("ADN_LSTM_NO_CLUSTER", raising=False)
        monkeypatch.delenv("ADN_LSTM_WIDE_PERSISTENT", raising=False)
        if mode == "single":
            monkeypatch.setenv("ADN_LSTM_NO_CLUSTER", "1")
            monkeypatch.setenv("ADN_LSTM_WIDE_PERSISTENT", "1")
        m = AdeNetModel(spec)
        m.set_params_dict(p)
        probs = m.predict(inputs, mask, 2)
        loss = m.compute_grads(inputs, y, mask, 2)
        out[mode] = (probs, loss, m.get_grads_dict())
        m.close()
    monkeypatch.delenv("ADN_LSTM_NO_CLUSTER", raising=False)
    monkeypatch.delenv("ADN_LSTM_WIDE_PERSISTENT", raising=False)
    valid = mask[..., None].astype(bool)
    np.testing.assert_array_equal(out["cluster"][0] * valid, out["single"][0] * valid)
    for k, g in out["cluster"][2].items():
        ref = out["single"][2][k]
        assert np.isfinite(g).all(), k
        assert np.abs(g - ref).max() <= 5e-3 * max(np.abs(ref).max(), 1e-6), k


@pytest.mark.parametrize("H,B,T,fusion", [(300, 33, 12, "concat"), (250, 63, 11, "concat"), (512, 100, 11, "adasum")])
def test_evaluation_is_reproducible_bit_for_bit(torch_cuda, lib, H, B, T, fusion):
    """bf16 mode: two evaluations of one batch give the same bits -- no forward-pass GEMM splits K (float atomics add in
    arrival order; these shapes used to differ by 1e-7 .. 1e-5 between calls), the LSTM kernels are deterministic."""
    from ip_avsr_amd.model import AdeNetModel
    spec = dict(O.spec_nstream([12, 9], enc_shapes=(14, 6), enc_acts=("rectify", "linear"), lstm_size=H, classes=5,
                               fusion=fusion, peepholes=False), precision="bf16")
    p, inputs, y, mask = make_case(spec, B, T, seed=7, perturb=0.02)
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    first = m.predict(inputs, mask, 2)
    for _ in range(4):
        np.testing.assert_array_equal(m.predict(inputs, mask, 2), first)
    l0 = m.loss(inputs, y, mask, 2)
    assert all(m.loss(inputs, y, mask, 2) == l0 for _ in range(3))
    m.close()


@pytest.mark.parametrize("N", [4052, 4056, 3844])
def test_pingpong_gemm_bf16_output_edges(torch_cuda, lib, N):
    """The ping-pong kernel's bf16 copy leaves in 16-byte stores after a lane exchange (a lane then holds 8 consecutive columns);
    at the right edge a lane may hold 4 valid columns -- its own or its partner's.  Shapes one launch of that kernel takes alone
    (16 x 16 tiles = one round of 256 CUs): N = 4052 ends 4 columns into a lane pair (the partner's half is what remains: a
    first version dropped exactly those), 4056 ends on the pair, 3844 leaves most of the last tile column empty.  Every element of the
    fp32 result and of the bf16 copy against the product of the bf16-rounded operands; pads untouched; M edge included."""
    torch = torch_cuda
    from ip_avsr_amd import _lib as L
    M, K = 4090, 256
    rng = np.random.default_rng(N)
    A = rng.normal(size=(M, K)).astype(np.float32); Bm = rng.normal(size=(K, N)).astype(np.float32)
    ld = (N + 63) // 64 * 64
    a_d = torch.tensor(A, device="cuda")
    b_d = torch.zeros((K, ld), device="cuda"); b_d[:, :N] = torch.tensor(Bm, device="cuda")
    a16, b16 = a_d.bfloat16().contiguous(), b_d.bfloat16().contiguous()
    ref = (a16.float() @ b16.float()[:, :N]).cpu().numpy()
    c_d = torch.full((M + 3, ld), 7.0, device="cuda")
    c16 = torch.full((M + 3, ld), 3.0, device="cuda", dtype=torch.bfloat16)
    L.check(lib.adn_op_gemm_shadow(0, M, N, K, dptr(a_d), K, dptr(b_d), ld, dptr(c_d), ld, dptr(a16), dptr(b16), dptr(c16), 0, None))
    torch.cuda.synchronize()
    out, out16 = c_d.cpu().numpy(), c16.float().cpu().numpy()
    assert np.abs(out[:M, :N] - ref).max() <= 2e-4
    assert (out[:M, N:] == 7.0).all() and (out[M:] == 7.0).all()
    want16 = torch.tensor(out[:M, :N]).bfloat16().float().numpy()
    np.testing.assert_array_equal(out16[:M, :N], want16)                 # the bf16 copy is the rounding of the fp32 result, everywhere
    assert (out16[:M, N:] == 3.0).all() and (out16[M:] == 3.0).all()


@pytest.mark.parametrize("M,N,K", [(40001, 300, 152), (33000, 2500, 150), (70000, 260, 25), (36000, 1000, 256)])
def test_a_stationary_nt_gemm_against_the_fp64_product(torch_cuda, lib, M, N, K):
    """gemm_bf16_nt_astat_kernel (csrc/gemm_bf16.hip: short K, bf16-only output, very many rows -- the conv auto-encoder's
    patch-gradient products): C16 = A16 . B16^T against the fp64 product of the bf16 operands, rounded to bfloat16; M not a
    multiple of the 64 / 128-row workgroups, N not a multiple of the 64-column chunks nor of 4 x 16, K with a masked tail inside
    an 8-element piece (150), K = 25 (one k-step), K = 256 (the limit); operand pad columns full of 1e30, output pad columns and the
    rows behind M untouched.  (ADN_GEMM_NO_ASTAT=1 sends the same call to the register-staged kernel.)"""
    torch = torch_cuda
    from ip_avsr_amd import _lib as L
    rng = np.random.default_rng(M + N + K)
    lda = (K + 7) // 8 * 8
    ldc = (N + 7) // 8 * 8 + 8
    A = np.zeros((M, lda), np.float32); A[:, :K] = rng.normal(size=(M, K))
    Bm = np.zeros((N, lda), np.float32); Bm[:, :K] = rng.normal(size=(N, K))
    if lda > K:                                       # what sits in the pad columns must not matter
        A[:, K:] = 1e30; Bm[:, K:] = -1e30
    a16 = torch.tensor(A, device="cuda").bfloat16().contiguous()
    b16 = torch.tensor(Bm, device="cuda").bfloat16().contiguous()
    ref = a16[:, :K].double().cpu().numpy() @ b16[:, :K].double().cpu().numpy().T
    want = torch.tensor(ref).bfloat16().float().numpy()
    c16 = torch.full((M + 2, ldc), 3.0, device="cuda", dtype=torch.bfloat16)
    L.check(lib.adn_op_gemm_shadow(1, M, N, K, dptr(a16), lda, dptr(b16), lda, None, ldc, dptr(a16), dptr(b16), dptr(c16), 0, None))
    torch.cuda.synchronize()
    out = c16.float().cpu().numpy()
    # one bfloat16 ulp of slack: fp32 accumulation order against the fp64 product
    assert (np.abs(out[:M, :N] - want) <= np.abs(want) * 2.0 ** -7 + 1e-2).all()
    assert (out[:M, N:] == 3.0).all() and (out[M:] == 3.0).all()
