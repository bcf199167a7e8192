"""oracle/rbm_oracle.py (the NumPy restatement of dbn/trainRBM.m & co.) on properties, since MATLAB is absent: the CD-1
statistics against a direct evaluation, the learning-rate / momentum schedule, the noise generators, the unfolded
auto-encoder; and the host side of ip_avsr_amd/dbn.py that needs no GPU."""
import numpy as np
import pytest

from oracle import rbm_oracle as R


def toy(rng, n=300, d=24):
    protos = rng.uniform(0, 1, (4, d)) > 0.5
    return (protos[rng.integers(0, 4, n)] ^ (rng.uniform(size=(n, d)) < 0.05)).astype(np.float64)


def test_noise_generators():
    g = dict(seed=7, counter=3)
    u = R.uniform(g, 0, (50000,))
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    z = R.normal(g, 0, (100000,), np.float64)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and np.isfinite(z).all()
    assert not np.array_equal(R.uniform(dict(seed=7, counter=4), 0, (100,)), R.uniform(g, 0, (100,)))


@pytest.mark.parametrize("layer_type,cd", [(("sigm", "sigm"), 1), (("sigm", "sigm"), 2), (("sigm", "ReLu"), 1), (("linear", "sigm"), 2)])
def test_one_batch_is_the_textbook_cd1_update(layer_type, cd):
    rng = np.random.default_rng(1)
    data = toy(rng, 100)
    p = R.dbn_params_init(1, [layer_type[1]], [10])
    p["rbmParams"]["type"] = cd
    rbm = R.init_rbm(24, 10, layer_type[0], layer_type[1], rng)
    rbm["hidbiases"] += 0.1; rbm["visbiases"] -= 0.05
    before = {k: v.copy() for k, v in rbm.items()}
    g = dict(seed=5, counter=0)
    err = R.cd1_batch(rbm, data, p, layer_type, 0.5, g)
    # direct evaluation
    W, hb, vb = before["W"], before["hidbiases"], before["visbiases"]
    hp, hs = R.rbm_up(data, W, hb, layer_type[1], g)
    vp, vs = R.rbm_down(hs, W, vb, layer_type[0], g if cd == 2 else None)
    v = vp if cd == 1 else vs
    hp2, _ = R.rbm_up(v, W, hb, layer_type[1])
    h = hp if cd == 1 else hs
    lrW, lrVb, lrHb = R.learning_rates(p, *layer_type)
    assert lrW == (0.001 if layer_type != ("sigm", "sigm") else 0.1)
    dW = lrW * ((data.T @ h - v.T @ hp2) / 100 - 0.0002 * W)
    np.testing.assert_allclose(rbm["W"], W + dW, atol=1e-12)
    np.testing.assert_allclose(rbm["hidbiases"], hb + lrHb * (h.sum(0) - hp2.sum(0)) / 100, atol=1e-12)
    np.testing.assert_allclose(rbm["visbiases"], vb + lrVb * (data.sum(0) - v.sum(0)) / 100, atol=1e-12)
    assert abs(err - ((data - v) ** 2).sum()) < 1e-9
    if layer_type[1] == "sigm":
        assert set(np.unique(hs)) <= {0.0, 1.0}
    # a second batch carries the momentum term
    w1 = rbm["W"].copy(); d1 = rbm["dW"].copy()
    R.cd1_batch(rbm, data, p, layer_type, 0.9, dict(seed=5, counter=1))
    assert np.abs(rbm["dW"] - 0.9 * d1).max() < np.abs(d1).max() * 2 and not np.array_equal(rbm["W"], w1)


def test_training_lowers_the_reconstruction_error_and_unfolds():
    rng = np.random.default_rng(0)
    data = toy(rng, 400, 20)
    p = R.dbn_params_init(1, ["sigm", "sigm"], [12, 6])
    p["rbmParams"]["epochs"] = 8
    rbm, eb, es = R.train_rbm(data, p, 12, ("sigm", "sigm"), rng)
    assert eb[-1] < 0.7 * eb[0] and len(eb) == 8 and abs(es[0] * 400 - eb[0] * 4) < 1e-9
    h = R.rbm_up(data, rbm["W"], rbm["hidbiases"], "sigm")[0]
    rbm2, _, _ = R.train_rbm(h, p, 6, ("sigm", "sigm"), rng)
    dbn = dict(W=[rbm["W"], rbm2["W"]], hidbiases=[rbm["hidbiases"], rbm2["hidbiases"]], visbiases=[rbm["visbiases"], rbm2["visbiases"]])
    weights, biases, acts, layers = R.unfold_dbn_to_ae(p, dbn, 20)
    assert [w.shape for w in weights] == [(20, 12), (12, 6), (6, 12), (12, 20)] and layers == [12, 6, 12, 20]
    assert acts == ["sigm", "sigm", "sigm", "sigm"]
    np.testing.assert_array_equal(weights[2], rbm2["W"].T)
    np.testing.assert_array_equal(biases[3], rbm["visbiases"])
    # the unfolded network computes RBMup, RBMup, RBMdown, RBMdown
    x = data[:5]
    a = x
    for w, b, t in zip(weights, biases, acts):
        a = R.compute_activations(t, a @ w + b)
    h2 = R.rbm_up(R.rbm_up(x, rbm["W"], rbm["hidbiases"], "sigm")[0], rbm2["W"], rbm2["hidbiases"], "sigm")[0]
    back = R.rbm_down(R.rbm_down(h2, rbm2["W"], rbm2["visbiases"], "sigm")[0], rbm["W"], rbm["visbiases"], "sigm")[0]
    np.testing.assert_allclose(a, back, atol=1e-12)
    with pytest.raises(ValueError):
        R.unfold_dbn_to_ae(p, dbn, 21)


def test_host_side_of_the_package_mirrors_the_oracle(tmp_path):
    from ip_avsr_amd import dbn as D
    from ip_avsr_amd.runners.nstream import load_decoder
    assert D.dbnParamsInit(1, ["ReLu"], [50]) == R.dbn_params_init(1, ["ReLu"], [50])
    rng = np.random.default_rng(3)
    dbn = dict(W=[rng.normal(size=(9, 5)).astype(np.float32), rng.normal(size=(5, 3)).astype(np.float32)],
               hidbiases=[rng.normal(size=5).astype(np.float32), rng.normal(size=3).astype(np.float32)],
               visbiases=[rng.normal(size=9).astype(np.float32), rng.normal(size=5).astype(np.float32)])
    p = D.dbnParamsInit(1, ["ReLu", "linear"], [5, 3])
    got = D.unfoldDBNtoAE(p, dbn, 9)
    want = R.unfold_dbn_to_ae(p, dbn, 9)
    for a, b in zip(got[0] + got[1], want[0] + want[1]):
        np.testing.assert_array_equal(a, b)
    assert got[2] == want[2] == ["ReLu", "linear", "ReLu", "sigm"] and got[3] == want[3]
    x = rng.normal(size=(4, 9)).astype(np.float32)
    np.testing.assert_allclose(D.RBMup(x, dbn["W"][0], dbn["hidbiases"][0], "ReLu"), R.rbm_up(x, dbn["W"][0], dbn["hidbiases"][0], "ReLu")[0], atol=1e-6)
    path = str(tmp_path / "ae.mat")
    D.save_ae_mat(path, got[0], got[1])
    w, b, shapes, _ = load_decoder(path, "5,3,5,9", "rectify,linear,rectify,sigmoid")     # what the runners read back
    assert shapes == [5, 3, 5, 9] and [x_.shape for x_ in w] == [(9, 5), (5, 3), (3, 5), (5, 9)]
    np.testing.assert_array_equal(b[3], dbn["visbiases"][0])
