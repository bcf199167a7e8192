"""Host-side driver pieces that need no GPU: evaluate_model2 (majority vote + confusion matrix),
load_decoder (.mat -> ae tuple), the confusion-matrix table, initialisers, checkpoint pickles."""
import os
import pickle

import numpy as np
import scipy.io as sio

from ip_avsr_amd import init as las_init
from ip_avsr_amd.runners.nstream import evaluate_model2, load_decoder
from ip_avsr_amd.utils import io as uio
from ip_avsr_amd.utils.plotting_utils import plot_confusion_matrix


def test_evaluate_model2_majority_vote_and_confusion():
    # hand-made probabilities, 3 utterances, T=4, 3 classes (pin with hand-made probs)
    probs = np.zeros((3, 4, 3))
    for (b, t, c) in [(0, 0, 2), (0, 1, 1), (0, 2, 2), (0, 3, 1),      # tie 1 vs 2 over 4 frames -> lowest id 1
                      (1, 0, 0), (1, 1, 2), (1, 2, 2), (1, 3, 2),      # only 2 valid frames: 0 and 2 tie -> 0
                      (2, 0, 1), (2, 1, 1), (2, 2, 0), (2, 3, 0)]:     # 3 valid frames -> 1
        probs[b, t, c] = 1.0
    mask = np.array([[1, 1, 1, 1], [1, 1, 0, 0], [1, 1, 1, 0]], np.uint8)
    y = np.array([1, 2, 1])
    seen = {}

    def eval_fn(x1, x2, m, w):
        seen["args"] = (x1, x2, m, w)
        return probs
    cr, conf = evaluate_model2(["s1", "s2"], y, mask, 9, eval_fn)
    assert seen["args"][0] == "s1" and seen["args"][3] == 9
    assert abs(cr - 2.0 / 3.0) < 1e-12
    want = np.zeros((3, 3), int); want[1, 1] = 2; want[2, 0] = 1
    np.testing.assert_array_equal(conf, want)
    from oracle import adenet_oracle as O
    np.testing.assert_array_equal(O.majority_vote(probs, mask), [1, 0, 1])


def test_load_decoder_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    dims = [12, 8, 6, 4, 3]
    d = {}
    for i in range(4):
        d["w%d" % (i + 1)] = rng.normal(size=(dims[i], dims[i + 1]))
        d["b%d" % (i + 1)] = rng.normal(size=(1, dims[i + 1]))
    for i in range(4, 8):                                   # decoder half, ignored (App. C)
        d["w%d" % (i + 1)] = rng.normal(size=(3, 3)); d["b%d" % (i + 1)] = rng.normal(size=(1, 3))
    path = str(tmp_path / "ae.mat")
    sio.savemat(path, d)
    w, b, shapes, nonlins = load_decoder(path, "8,6,4,3", "rectify,rectify,rectify,linear")
    assert shapes == [8, 6, 4, 3] and nonlins == ["rectify", "rectify", "rectify", "linear"]
    assert len(w) == 4 and w[0].dtype == np.float32 and w[0].shape == (12, 8) and b[3].shape == (3,)
    np.testing.assert_allclose(w[2], d["w3"].astype("float32"))
    np.testing.assert_allclose(b[1], d["b2"][0].astype("float32"))


def test_confusion_table_and_split_file(tmp_path):
    conf = np.array([[2, 0], [1, 3]])
    table = plot_confusion_matrix(conf, ["a", "b"], fmt="pipe")
    assert "| a" in table and "3" in table.splitlines()[-1]
    p = tmp_path / "subjects.txt"
    p.write_text("1,5,9\n")
    assert uio.read_data_split_file(str(p)) == [1, 5, 9]


def test_param_pickle_is_a_plain_list_of_arrays(tmp_path):
    class Net:
        def __init__(self):
            self.v = [np.arange(6, dtype="float32").reshape(2, 3), np.zeros(3, "float32")]

        def get_all_param_values(self):
            return [a.copy() for a in self.v]

        def set_all_param_values(self, vals):
            self.v = [np.asarray(a) for a in vals]
    net = Net()
    path = str(tmp_path / "best.pkl")
    uio.save_model_params(net, path)
    raw = pickle.load(open(path, "rb"))
    assert isinstance(raw, list) and raw[0].shape == (2, 3)            # utils/io.py:40-42 format
    net.v = [a * 0 for a in net.v]
    uio.load_model_params(net, path)
    np.testing.assert_array_equal(net.v[0], np.arange(6).reshape(2, 3))


def test_initialisers():
    las_init.set_rng(np.random.RandomState(0))
    g = las_init.GlorotUniform()((150, 250))
    assert g.dtype == np.float32 and abs(g).max() <= np.sqrt(6.0 / 400) + 1e-7
    o = las_init.Orthogonal()((30, 50))
    np.testing.assert_allclose(o @ o.T, np.eye(30), atol=1e-5)
    o2 = las_init.Orthogonal()((50, 30))
    np.testing.assert_allclose(o2.T @ o2, np.eye(30), atol=1e-5)
    assert abs(las_init.Uniform()((1000,))).max() <= 0.01
    assert 0.08 < las_init.Normal(0.1)((4000,)).std() < 0.12
    assert isinstance(las_init.select("ortho"), las_init.Orthogonal)
    assert isinstance(las_init.select("anything-else"), las_init.GlorotUniform)
    assert isinstance(las_init.resolve(las_init.GlorotUniform), las_init.GlorotUniform)


class _FakeNetwork(object):
    """Stands in for AdeNetModel in runners/nstream.fit (host_batches=True: the reference's generators, no GPU): records what
    the loop hands over and returns costs that improve for three epochs, then get worse."""

    def __init__(self, n_streams, classes):
        self.S, self.C, self.head, self.spec = n_streams, classes, "frames", {"streams": [{} for _ in range(n_streams)]}
        self.steps, self.evals, self.snapshots = [], 0, 0

    def compile(self, lr, order):
        def cost(*a):
            self.evals += 1
            e = self.evals // 2                        # (train cost + validation cost per epoch)
            return np.float32(2.0 - 0.1 * e if e <= 3 else 1.7 + 0.05 * (e - 3))
        def val_fn(*a):
            mask = a[1] if self.S == 2 else a[self.S]          # (the 2-stream runner compiles val_fn(in1, mask, in2, window))
            probs = np.zeros(mask.shape + (self.C,), np.float32); probs[..., 1] = 1.0
            return probs
        return None, cost, cost, val_fn

    def train_step(self, Xs, y, m, w, lr, want_loss=True):
        assert not want_loss                           # the loop must not wait for the cost the reference discards
        self.steps.append((len(Xs), Xs[0].shape, y.shape, y.dtype, m.dtype, int(m.sum()), w, lr, [x[:, 0, 0].copy() for x in Xs]))

    def synchronize(self):
        pass

    def snapshot_params(self):
        self.snapshots += 1
        return ("snapshot", self.snapshots)


def test_epoch_loop_on_the_host_generators():
    """runners/nstream.fit with the reference's host-side generators (ADN_HOST_BATCHES path) and a recording network: batch
    shapes and dtypes as runners/3stream.py:357-370 builds them, the utterance order of gen_lstm_batch_random from the global
    NumPy stream (after the two held-out draws), best-parameter snapshots only on improvement, early stop by early_stop2."""
    from ip_avsr_amd.runners import nstream
    from ip_avsr_amd.utils import datagen as dg
    rng = np.random.RandomState(3)
    lens = {k: rng.randint(3, 9, size=n) for k, n in (("train", 11), ("val", 5), ("test", 4))}
    split = {k: [rng.normal(size=(int(l.sum()), d)).astype(np.float32) for d in (6, 4)] for k, l in lens.items()}
    ys = {k: np.repeat(np.arange(len(l)) % 3, l) for k, l in lens.items()}
    net = _FakeNetwork(2, 3)
    np.random.seed(21)
    st = nstream.fit(net, split, ys, lens, 2, windowsize=3, num_epoch=9, epochsize=4, batchsize=4, validation_window=2, learning_rate=0.5,
                     say=lambda *a, **k: None, host_batches=True, progress=False)
    assert len(net.steps) == 4 * len(st["cost_val"]) and len(st["train_seconds"]) == len(st["epoch_seconds"]) == len(st["cost_val"])
    assert 4 <= len(st["cost_val"]) < 9                                           # stopped early once the cost kept rising
    assert net.snapshots == 3 and st["best_params"] == ("snapshot", 3)            # three improving epochs
    assert st["best_val"] == min(st["cost_val"])
    # the same draws in the same order: held-out splits first (two permutations each), then the training stream
    np.random.seed(21)
    for k in ("val", "test"):
        next(dg.gen_lstm_batch_random(split[k][0], ys[k], lens[k], batchsize=len(lens[k])))
    ref = dg.gen_lstm_batch_random(split["train"][0], ys["train"], lens["train"], batchsize=4)
    tmax = int(lens["train"].max())
    for n_in, shape, yshape, ydt, mdt, frames, w, lr, firsts in net.steps[:6]:
        X1, yb, mb, ib = next(ref)
        assert n_in == 2 and shape == X1.shape and yshape == (len(ib), tmax) and ydt == np.uint8 and mdt == np.uint8
        assert frames == int(mb.sum()) and w == 3 and lr == 0.5
        assert np.array_equal(firsts[0], X1[:, 0, 0])
    assert [s[1][0] for s in net.steps[:3]] == [4, 4, 3]                          # 11 utterances: 4, 4, 3 (short), reshuffle


def test_avletters_scripts_pick_their_network_from_has_encoder(monkeypatch):
    """reference avletters/{1,2,3}stream.py read a ``has_encoder`` option per stream (schema 1 of SURVEY App. B as those
    scripts use it): an encoder-less single stream is deltanet_v1 (:213-238), an encoder-less SECOND stream of the 2-stream
    script is adenet_v2 (:265-286); the 3-stream script only builds a network when stream 2 has an encoder (:278) and needs
    ae1 / ae3 in every branch -- there this package raises instead of the reference's NameError."""
    import pytest
    from ip_avsr_amd.modelzoo import _factory as F
    from ip_avsr_amd.runners.nstream import build_network_avletters
    monkeypatch.setattr(F, "SPEC_ONLY", True)
    rng = np.random.default_rng(0)
    dims = [24, 16, 12, 8, 5]
    loaded = []

    def load_ae(k):
        loaded.append(k)
        return ([rng.normal(size=(a, b)).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])],
                [np.zeros(b, np.float32) for b in dims[1:]], dims[1:], ["rectify", "rectify", "rectify", "linear"])
    cfg = dict(weight_init_fn="glorot", use_peepholes=False, lstm_size=6, output_classes=4, fusiontype="sum", use_blstm=True,
               use_blstm_substream=False)
    spec = build_network_avletters(1, [False], load_ae, [24], [None], cfg)[0]
    assert loaded == [] and spec["streams"][0]["enc_names"] == [] and spec["streams"][0]["delta"]
    assert spec["streams"][0]["lstm_names"] == ["f_lstm", "b_lstm"]
    spec = build_network_avletters(1, [True], load_ae, [24], [None], cfg)[0]
    assert loaded == [0] and len(spec["streams"][0]["enc_names"]) == 4
    del loaded[:]
    spec, _ = build_network_avletters(2, [True, False], load_ae, [24, 9], [None, None], cfg)
    assert loaded == [0]                                     # stream 2's encoder file is never opened (avletters/2stream.py:248)
    assert len(spec["streams"][0]["enc_names"]) == 4 and spec["streams"][1]["enc_names"] == [] and spec["streams"][1]["input_dim"] == 9
    spec, _ = build_network_avletters(2, [True, True], load_ae, [24, 24], [None, None], cfg)
    assert all(len(s["enc_names"]) == 4 for s in spec["streams"])
    spec, _ = build_network_avletters(3, [True, True, True], load_ae, [24, 24, 24], [None, None, None], dict(cfg, fusiontype="concat"))
    assert len(spec["streams"]) == 3 and spec["fusion"] == "concat"
    with pytest.raises(ValueError, match="stream2"):
        build_network_avletters(3, [True, False, True], load_ae, [24, 9, 24], [None, None, None], cfg)
    with pytest.raises(ValueError, match="ae3"):
        build_network_avletters(3, [True, True, False], load_ae, [24, 24, 9], [None, None, None], cfg)
    with pytest.raises(ValueError, match="ae1"):
        build_network_avletters(2, [False, True], load_ae, [24, 24], [None, None], cfg)


def test_plane_input_is_the_hi_lo_split_of_a_float32_tensor():
    """model.PlaneInput (the resident form of the bf16x3 / mixed arithmetic, ADN_FLAG_PLANE_INPUTS): hi = bf16(x), lo = bf16(x - hi);
    hi + lo recovers x to 2^-16 of its magnitude (a 16-bit significand), indexing slices both planes alike, and the constructor
    refuses anything but two bfloat16 tensors of one shape.  Host logic only: CPU tensors."""
    import pytest
    import torch
    from ip_avsr_amd.model import PlaneInput
    g = torch.Generator().manual_seed(5)
    x = torch.randn(7, 11, 24, generator=g) * torch.tensor([1e-3, 1.0, 300.0]).repeat(8)
    p = PlaneInput.split(x)
    assert p.hi.dtype == torch.bfloat16 and p.lo.dtype == torch.bfloat16 and tuple(p.shape) == (7, 11, 24) and p.ndim == 3 and len(p) == 7
    assert torch.equal(p.hi, x.to(torch.bfloat16))
    assert torch.equal(p.lo, (x - p.hi.float()).to(torch.bfloat16))
    err = (p.float() - x).abs()
    assert bool((err <= x.abs() * 2.0 ** -16 + 1e-30).all())
    q = p[2:5]
    assert torch.equal(q.hi, p.hi[2:5]) and torch.equal(q.lo, p.lo[2:5]) and q.hi.is_contiguous()
    with pytest.raises(ValueError):
        PlaneInput(p.hi, p.lo[:3])
    with pytest.raises(ValueError):
        PlaneInput(p.hi.float(), p.lo)
