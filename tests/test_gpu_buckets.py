"""Length buckets of the recurrent side (include/adenet.h adn_set_length_buckets; csrc/model.hip TmPlan): a compacted train step keeps
its time-major tensors as 2 - 4 buckets of equally many utterances, each as long as its longest one, instead of B x T rows.  A padding
frame contributes nothing to the loss or to any gradient, so the bucketed step must give the unbucketed step's loss and gradients up
to the order of the sums -- checked per arithmetic at the tests' size and at the bench geometry, against the fp64 oracle, over a
sequence of batches with different lengths (the tables, the spare blocks and the exchange buffers are re-made under a running
model), and for the cases that must stay on the B x T layout."""
import os

import numpy as np
import pytest

from oracle import adenet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


@pytest.fixture(autouse=True)
def small_batches_compact_too(monkeypatch):
    """(as tests/test_gpu_compact.py: the library leaves batches under 8192 rows padded -- and a padded call never buckets)"""
    monkeypatch.setenv("ADN_COMPACT_MIN_ROWS", "0")
    monkeypatch.setenv("ADN_CHECK_PADDING", "1")


def _declined(prec):
    """the diagnostic switches of profiles/scripts/envmatrix.sh under which a train step keeps the B x T layout by design"""
    return bool(os.environ.get("ADN_NO_COMPACT") or os.environ.get("ADN_STREAMS") or os.environ.get("ADN_NO_LENGTH_BUCKETS") or
                os.environ.get("ADN_DETERMINISTIC") or os.environ.get("ADN_LSTM_NO_CLUSTER") or
                (prec == "bf16" and os.environ.get("ADN_BF16_NO_SHADOW")) or
                (prec in ("bf16x3", "mixed") and (os.environ.get("ADN_X3_NO_PLANES") or os.environ.get("ADN_LSTM_NO_X3_CLUSTER") or
                                                  os.environ.get("ADN_LSTM_NO_X3_CLUSTER_BWD"))) or
                (prec == "bf16" and os.environ.get("ADN_LSTM_NO_CLUSTER_BWD")))       # (mixed: the bf16x3 backward kernel steps in)


def _data(spec, B, T, dims, seed, perturb=0.05, lo=None):
    rng = np.random.default_rng(seed)
    p = O.init_params(spec, rng, np.float32, enc_std=0.1, perturb=perturb)
    lens = rng.integers(lo if lo else max(2, T // 3), T + 1, size=B)
    lens[rng.integers(0, B)] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    xs = [(rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32) for d in dims]
    y = rng.integers(0, 26, size=(B, 1)).repeat(T, axis=1).astype(np.int32)
    return p, lens.astype(np.int32), mask, xs, y


def _expected_rows(lens, B, nb):
    """the plan's row count for nb buckets (model.hip setup_buckets): equal cuts of the length-sorted batch, one spare block between"""
    Bb = -(-B // nb)
    order = np.sort(lens)[::-1]
    tk = [int(order[k * Bb]) if k * Bb < B else 1 for k in range(nb)]
    return Bb * (sum(t + 1 for t in tk) - 1)


def _rel(a, b, scale):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3 * scale))


def _both_layouts(m, lens, xs, y, mask, theta):
    out = {}
    for mode in ("bxt", "buckets"):
        m.set_length_buckets(mode == "buckets")
        m.set_batch_lengths(lens)
        loss = m.compute_grads(xs, y, mask, theta)
        out[mode] = (loss, m.get_grads_dict(), m.bucket_rows(), m.compact_rows())
    return out


@pytest.mark.parametrize("prec", ["bf16x3", "mixed", "bf16"])
@pytest.mark.parametrize("fusion", ["concat", "sum"])
def test_bucketed_step_equals_the_step_over_b_x_t_rows(torch_cuda, prec, fusion):
    from ip_avsr_amd.model import AdeNetModel
    dims = (72, 56)
    spec = O.spec_nstream(list(dims), enc_shapes=(160, 128, 50), enc_acts=("rectify", "rectify", "linear"), lstm_size=40, classes=26,
                          fusion=fusion)
    B, T, theta = 70, 30, 9
    p, lens, mask, xs, y = _data(spec, B, T, dims, 8)
    m = AdeNetModel(dict(spec, precision=prec))
    m.set_auto_compaction(False)
    m.set_params_dict(p)
    out = _both_layouts(m, lens, xs, y, mask, theta)
    m.close()
    assert out["bxt"][2] == 0
    if _declined(prec):
        assert out["buckets"][2] == 0
        return
    rows = out["buckets"][2]
    assert rows in [_expected_rows(lens, B, nb) for nb in (2, 3, 4)] and rows <= 0.9 * B * T, rows
    assert out["buckets"][3] == out["bxt"][3] == int(lens.sum()) + 1
    # the same products, summed in another order (other row counts: other tiles, other K-slices): fp32 rounding of the accumulations
    assert abs(out["buckets"][0] - out["bxt"][0]) <= 2e-6 * abs(out["bxt"][0])
    gscale = max(np.abs(v).max() for v in out["bxt"][1].values())
    errs = {k: _rel(out["buckets"][1][k], out["bxt"][1][k], gscale) for k in O.param_names(spec)}
    print("buckets vs B x T, %s / %s: %d rows instead of %d, worst gradient difference %.2e of its scale (%s)"
          % (prec, fusion, rows, B * T, max(errs.values()), max(errs, key=errs.get)))
    for k, e in errs.items():
        assert e <= 2e-4, (k, e)


def test_bucketed_bf16x3_gradients_against_the_oracle(torch_cuda):
    from ip_avsr_amd.model import AdeNetModel
    dims = (72, 56, 64)
    spec = O.spec_nstream(list(dims), enc_shapes=(96, 64, 24), enc_acts=("rectify", "rectify", "linear"), lstm_size=40, classes=26,
                          fusion="concat")
    B, T, theta = 44, 24, 9
    p, lens, mask, xs, y = _data(spec, B, T, dims, 12)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    l_ref, g_ref, _ = O.loss_and_grads(spec, p64, [x.astype(np.float64) for x in xs], y, mask, theta)
    m = AdeNetModel(dict(spec, precision="bf16x3"))
    m.set_auto_compaction(False)
    m.set_params_dict(p)
    m.set_batch_lengths(lens)
    loss = m.compute_grads(xs, y, mask, theta)
    g = m.get_grads_dict()
    rows = m.bucket_rows()
    m.close()
    assert (rows == 0) == _declined("bf16x3")
    assert abs(loss - l_ref) <= 2e-6 * abs(l_ref)
    gscale = max(np.abs(v).max() for v in g_ref.values())
    for k in O.param_names(spec):
        assert _rel(g[k], g_ref[k], gscale) <= 2e-4, k


@pytest.mark.parametrize("prec", ["bf16", "bf16x3", "mixed"])
def test_a_sequence_of_batches_with_other_lengths_each(torch_cuda, prec):
    """every call re-cuts the buckets or leaves them: new tables, another bucket size or none (the regions of the LSTM exchange
    buffers move: what was a forward region becomes an inbox), spare blocks where an earlier call kept gradients, another (B, T) in
    between -- each call against a second model of the same parameters that never buckets, at the order-of-summation grade"""
    from ip_avsr_amd.model import AdeNetModel
    dims = (72, 56)
    spec = O.spec_nstream(list(dims), enc_shapes=(96, 64, 24), enc_acts=("rectify", "rectify", "linear"), lstm_size=40, classes=26,
                          fusion="concat")
    theta = 9
    p = O.init_params(spec, np.random.default_rng(3), np.float32, enc_std=0.1, perturb=0.05)
    models = []
    for on in (False, True):
        m = AdeNetModel(dict(spec, precision=prec))
        m.set_auto_compaction(False)
        m.set_length_buckets(on)
        m.set_params_dict(p)
        models.append(m)
    seen = []
    plan = [(70, 30, 4, 1), (70, 30, 20, 2), (70, 30, 4, 3), (70, 30, 29, 5), (150, 30, 4, 6), (150, 30, 28, 7), (96, 24, 3, 4), (70, 30, 4, 1),
            (70, 30, 4, 1), (70, 30, 25, 8), (70, 30, 6, 9)]
    for step, (B, T, lo, seed) in enumerate(plan):
        _, lens, mask, xs, y = _data(spec, B, T, dims, 100 + seed, lo=lo)
        res = []
        for m in models:
            m.set_batch_lengths(lens)
            loss = m.compute_grads(xs, y, mask, theta)
            res.append((loss, m.get_grads_dict(), m.bucket_rows()))
        seen.append(res[1][2])
        assert res[0][2] == 0
        cuts = [_expected_rows(lens, B, nb) for nb in (2, 3, 4)]
        if min(cuts) > 0.9 * B * T or _declined(prec):
            assert res[1][2] == 0                      # (nearly full utterances: under 10 % to save -- the call stays on B x T rows)
        elif os.environ.get("ADN_LSTM_CUS"):           # (a smaller device may not hold the finest cut's launches resident)
            assert res[1][2] == 0 or (res[1][2] in cuts and res[1][2] <= 0.9 * B * T)
        else:
            assert res[1][2] == min(cuts)
        assert abs(res[1][0] - res[0][0]) <= 2e-6 * abs(res[0][0]), (step, res[1][0], res[0][0])
        gscale = max(np.abs(v).max() for v in res[0][1].values())
        for k in O.param_names(spec):
            assert _rel(res[1][1][k], res[0][1][k], gscale) <= 2e-4, (step, k)
    for m in models:
        m.close()
    print("time-major rows of the calls:", seen)
    assert _declined(prec) or os.environ.get("ADN_LSTM_CUS") or (len(set(seen)) >= 5 and 0 in seen)      # several cuts, and the uncut layout in between


def test_forward_only_calls_and_the_probabilities_of_a_bucketed_step(torch_cuda):
    from ip_avsr_amd.model import AdeNetModel
    dims = (72, 56)
    spec = O.spec_nstream(list(dims), enc_shapes=(96, 64, 24), enc_acts=("rectify", "rectify", "linear"), lstm_size=40, classes=26,
                          fusion="concat")
    B, T, theta = 70, 30, 9
    p, lens, mask, xs, y = _data(spec, B, T, dims, 21)
    m = AdeNetModel(dict(spec, precision="bf16x3"))
    m.set_auto_compaction(False)
    m.set_params_dict(p)
    m.set_batch_lengths(lens)
    probs = m.predict(xs, mask, theta)
    assert m.bucket_rows() == 0 and probs.shape == (B, T, 26)            # val_fn's output has every frame: never bucketed
    m.set_batch_lengths(lens)
    loss, probs2 = m.loss_and_probs(xs, y, mask, theta)
    assert m.bucket_rows() == 0 and np.abs(probs2 - probs).max() <= 1e-6
    m.set_batch_lengths(lens)
    m.compute_grads(xs, y, mask, theta)
    if not _declined("bf16x3"):
        assert m.bucket_rows() > 0
        import ctypes as C
        out = np.empty((B, T, 26), np.float32)
        rc = m._lib.adn_read_probs(m._handle, B, T, 0, out.ctypes.data_as(C.c_void_p))
        assert rc != 0                                  # per-frame outputs of a bucketed step are refused, not returned half-filled
    m.compute_grads(xs, y, mask, theta)                 # no announcement: padded, unbucketed
    assert m.bucket_rows() == 0 and m.compact_rows() == 0
    m.close()


@pytest.mark.parametrize("what", ["dropout", "last_head", "deterministic"])
def test_models_that_walk_the_time_axis_themselves_stay_on_b_x_t_rows(torch_cuda, what):
    from ip_avsr_amd.model import AdeNetModel
    from ip_avsr_amd import _lib
    dims = (72, 56)
    spec = O.spec_nstream(list(dims), enc_shapes=(96, 64, 24), enc_acts=("rectify", "rectify", "linear"), lstm_size=40, classes=26,
                          fusion="concat")
    if what == "dropout":
        spec["agg_dropout"] = 0.2
    if what == "last_head":
        spec.update(head="last", loss="cross_entropy")
    B, T, theta = 70, 30, 9
    p, lens, mask, xs, y = _data(spec, B, T, dims, 5)
    lib = _lib.load()
    was = lib.adn_get_deterministic()
    if what == "deterministic":
        lib.adn_set_deterministic(1)
    try:
        m = AdeNetModel(dict(spec, precision="bf16"))
        m.set_auto_compaction(False)
        m.set_params_dict(p)
        m.set_batch_lengths(lens)
        m.compute_grads(xs, y, mask, theta)
        assert m.bucket_rows() == 0
        m.close()
    finally:
        lib.adn_set_deterministic(was)
