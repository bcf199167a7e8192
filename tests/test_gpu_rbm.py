"""RBM / DBN pre-trainer on the GPU (csrc/rbm.hip, ip_avsr_amd/dbn.py; reference dbn/trainRBM.m & co.) against
oracle/rbm_oracle.py with the same counter-based noise: minibatch-by-minibatch parity of weights, biases, momentum terms
and the reconstruction error for every layer-type pair the reference's scripts use, then the whole pre-training chain:
trainDBN -> unfoldDBNtoAE -> .mat -> the dense encoder of a stream."""
import numpy as np
import pytest

from oracle import rbm_oracle as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ip_avsr_amd import dbn
    return dbn


def images(rng, n, d):
    protos = rng.uniform(0, 1, (6, d))
    x = protos[rng.integers(0, 6, n)] + 0.1 * rng.normal(size=(n, d))
    return np.clip(x, 0, 1).astype(np.float32)


@pytest.mark.parametrize("layer_type,cd,dims,hid", [(("sigm", "sigm"), 1, 300, 120), (("sigm", "sigm"), 2, 77, 50),
                                                     (("sigm", "ReLu"), 1, 1200, 500), (("ReLu", "ReLu"), 1, 130, 64),
                                                     (("ReLu", "linear"), 1, 64, 50), (("linear", "sigm"), 2, 50, 30)])
def test_minibatches_match_the_oracle(D, layer_type, cd, dims, hid):
    rng = np.random.default_rng(dims + hid)
    p = D.dbnParamsInit(1, [layer_type[1]], [hid])
    p["rbmParams"]["type"] = cd
    data = images(rng, 260, dims)
    ref = R.init_rbm(dims, hid, layer_type[0], layer_type[1], rng, np.float64)
    ref["hidbiases"] += 0.05
    m = D.RBM(dims, hid, layer_type, p)
    m.set(0, ref["W"]); m.set(1, ref["hidbiases"]); m.set(2, ref["visbiases"])
    ref = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in ref.items()}
    for step, (lo, hi, mom) in enumerate([(0, 100, 0.5), (100, 200, 0.5), (200, 260, 0.9)]):       # incl. a short last batch
        err_ref = R.cd1_batch(ref, data[lo:hi].astype(np.float64), p, layer_type, mom, dict(seed=99, counter=step))
        err = m.train_batch(data[lo:hi], mom, 99, step)
        # a Bernoulli draw whose probability sits within fp32 rounding of its uniform may flip (expected < 0.1 per batch):
        # one flipped unit moves the error by <= 1 and a weight by <= lr / batchsize
        assert abs(err - err_ref) <= 2e-4 * err_ref + (2.0 if "sigm" in layer_type else 0.0), (step, err, err_ref)
        for which, key in enumerate(("W", "hidbiases", "visbiases", "dW", "dhid", "dvis")):
            got, want = m.get(which), ref[key]
            tol = 2e-5 * max(np.abs(want).max(), 1e-3) + (2.5e-3 if "sigm" in layer_type else 0.0) * (0.1 / 100)
            assert np.abs(got - want).max() <= tol, (step, key, np.abs(got - want).max(), tol)
    np.testing.assert_allclose(m.up(data[:7]), R.rbm_up(data[:7].astype(np.float64), ref["W"], ref["hidbiases"], layer_type[1])[0],
                               atol=2e-5)
    m.close()


def test_pretraining_chain_feeds_an_encoder(D, tmp_path):
    """exampleDBN_AE.m's flow at toy size: 144-d 'mouth images' -> DBN 60-30-12 -> unfolded auto-encoder -> .mat -> the
    encoder weights of a 1-stream model; the reconstruction error falls during pre-training."""
    from ip_avsr_amd.runners.nstream import load_decoder
    rng = np.random.RandomState(4)
    data = images(np.random.default_rng(4), 1000, 144)
    p = D.dbnParamsInit(1, ["ReLu", "ReLu", "linear"], [60, 30, 12])
    p["rbmParams"]["epochs"] = 6
    dbn, eb, es = D.trainDBN(data, p, rng=rng, seed=11, verbose=False)
    assert [w.shape for w in dbn["W"]] == [(144, 60), (60, 30), (30, 12)]
    assert all(np.isfinite(e).all() for e in eb) and eb[0][-1] < eb[0][0]
    weights, biases, acts, layers = D.unfoldDBNtoAE(p, dbn, 144)
    assert layers == [60, 30, 12, 30, 60, 144] and acts == ["ReLu", "ReLu", "linear", "ReLu", "ReLu", "sigm"]
    path = str(tmp_path / "pretrained.mat")
    D.save_ae_mat(path, weights, biases)
    w, b, shapes, nonlins = load_decoder(path, "60,30,12", "rectify,rectify,linear")
    from ip_avsr_amd.modelzoo import deltanet_majority_vote
    net = deltanet_majority_vote.create_model((w, b, shapes, nonlins), (None, None, 144), None, (None, None), None, lstm_size=20,
                                              win=3, output_classes=5)
    vals = net.get_all_param_values()                      # Lasagne order: fc1.W, fc1.b, fc2.W, fc2.b, bottleneck.W, bottleneck.b, ...
    np.testing.assert_array_equal(vals[0], dbn["W"][0])
    np.testing.assert_array_equal(vals[5], dbn["hidbiases"][2])
    net.close()
