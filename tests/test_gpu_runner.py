"""End-to-end run of the N-stream driver on an MI355X: synthetic .mat datasets + DBN .mat files + an .ini in
the reference's schema; checks the loop runs, the cost falls, the outputs (results line, pickle) appear and
the pickle reloads into an identical model."""
import os

import numpy as np
import pytest
import scipy.io as sio

pytestmark = pytest.mark.gpu

INI = """
[stream{k}]
data = {root}/s{k}.mat
imagesize = 4,6
model = {root}/ae{k}.mat
input_dimensions = 24
shape = 16,12,8,5
nonlinearities = rectify,rectify,rectify,linear
reorderdata = {reorder}
diffimage = {diff}
meanremove = True
samplewisenormalize = True
featurewisenormalize = False
force_align_data = False
"""

TAIL = """
[lstm_classifier]
fusiontype = {fusion}
weight_init = glorot
use_peepholes = False
windowsize = 3
output_classes = 4
output_classnames = a,b,c,d
lstm_size = 12
matlab_target_offset = True
use_dropout = {dropout}
use_blstm = True

[training]
validation_window = 4
num_epoch = 6
learning_rate = 0.01
epochsize = 5
batchsize = 8
train_subjects_file = {root}/train.txt
val_subjects_file = {root}/val.txt
test_subjects_file = {root}/test.txt
"""


def make_dataset(root, n_streams):
    rng = np.random.RandomState(0)
    subjects = np.repeat(np.arange(1, 9), 6)                 # 8 subjects x 6 utterances
    n = len(subjects)
    lens = rng.randint(5, 11, size=n)
    labels = np.arange(n) % 4 + 1                            # 1-based (matlab_target_offset)
    protos = rng.normal(size=(4, 24)) * 2
    for k in range(1, n_streams + 1):
        rows, tv = [], []
        for u in range(n):
            ramp = np.linspace(0, 1, lens[u])[:, None]
            rows.append(protos[labels[u] - 1][None, :] * (0.5 + ramp) + rng.normal(size=(lens[u], 24)) * 0.3)
            tv.append(np.full(lens[u], labels[u]))
        sio.savemat(os.path.join(root, "s%d.mat" % k),
                    dict(dataMatrix=np.concatenate(rows), targetsVec=np.concatenate(tv)[:, None].astype("uint8"),
                         videoLengthVec=lens[:, None], subjectsVec=subjects[:, None]))
        dims = [24, 16, 12, 8, 5]
        ae = {}
        for i in range(4):
            ae["w%d" % (i + 1)] = rng.normal(0, 0.3, (dims[i], dims[i + 1]))
            ae["b%d" % (i + 1)] = rng.normal(0, 0.05, (1, dims[i + 1]))
        sio.savemat(os.path.join(root, "ae%d.mat" % k), ae)
    open(os.path.join(root, "train.txt"), "w").write("1,2,3,4,5")
    open(os.path.join(root, "val.txt"), "w").write("6,7")
    open(os.path.join(root, "test.txt"), "w").write("8")


@pytest.mark.parametrize("n_streams,fusion,dropout", [(3, "concat", False), (1, "none", False), (2, "adasum", False),
                                                      (3, "sum", True)])
def test_runner_end_to_end(tmp_path, n_streams, fusion, dropout):
    from ip_avsr_amd.runners import nstream
    from ip_avsr_amd.utils.io import load_model
    root = str(tmp_path)
    make_dataset(root, n_streams)
    ini = "".join(INI.format(k=k, root=root, reorder=(k == 1), diff=(k == 2)) for k in range(1, n_streams + 1))
    ini += TAIL.format(root=root, fusion=fusion, dropout=dropout)
    cfg = os.path.join(root, "cfg.ini")
    open(cfg, "w").write(ini)
    res_file, best_file = os.path.join(root, "results.csv"), os.path.join(root, "best.pkl")
    out = nstream.main(n_streams, ["--config", cfg, "--write_results", res_file, "--save_best", best_file,
                                   "--seed", "7"])
    assert len(out["cost_val"]) >= 2 and np.isfinite(out["cost_val"]).all()
    assert min(out["cost_val"]) < out["cost_val"][0] or out["best_cr"] >= 0.5      # it learns something
    line = open(res_file).read().strip().split(",")
    assert len(line) == 3 and abs(float(line[1]) - out["best_cr"]) < 1e-9
    values = load_model(best_file)
    net = out["network"]
    if dropout:                                  # use_dropout picks adenet_3stream_dropout (runners/3stream.py:284-291)
        assert net.spec["agg_dropout"] == 0.5 and net.H == 24 and all(s["dropout"] == 0.5 for s in net.spec["streams"])
    assert isinstance(values, list) and len(values) == len(net.params)
    for v, p in zip(values, net.params):
        assert v.shape == p.shape and v.dtype == np.float32
    np.testing.assert_array_equal(values[0], net.get_all_param_values()[0])       # best params were restored
    net.close()


def test_avletters_two_stream_script_with_an_encoderless_stream(tmp_path):
    """reference avletters/2stream.py on an INI whose second stream has ``has_encoder = False`` (:244, :265-286: adenet_v2 --
    an encoder stream + a raw second stream): the iterVec train / val split (:230-231), no test split, the two-field
    results line (:415) and the best-epoch checkpoint."""
    from ip_avsr_amd.runners import nstream
    from ip_avsr_amd.utils.io import load_model
    root = str(tmp_path)
    make_dataset(root, 2)
    # the .mat files of the avletters scripts carry an iteration vector: utterances of iterations 1, 2 train, 3 validates
    for k in (1, 2):
        path = os.path.join(root, "s%d.mat" % k)
        d = {key: v for key, v in sio.loadmat(path).items() if not key.startswith("__")}
        d["iterVec"] = (np.arange(len(d["videoLengthVec"])) % 3 + 1)[:, None]
        sio.savemat(path, d)
    ini = "".join(INI.format(k=k, root=root, reorder=(k == 1), diff=False) + "has_encoder = %s\n" % (k == 1) for k in (1, 2))
    ini += TAIL.format(root=root, fusion="sum", dropout=False) + "use_blstm_substream = False\n"
    cfg = os.path.join(root, "cfg.ini")
    open(cfg, "w").write(ini)
    res_file, best_file = os.path.join(root, "results.csv"), os.path.join(root, "best.pkl")
    out = nstream.main(2, ["--config", cfg, "--write_results", res_file, "--save_best", best_file, "--seed", "7"], variant="avletters")
    net = out["network"]
    assert len(net.spec["streams"][0]["enc_names"]) == 4 and net.spec["streams"][1]["enc_names"] == []
    assert len(out["cost_val"]) >= 2 and np.isfinite(out["cost_val"]).all()
    assert min(out["cost_val"]) < out["cost_val"][0] or out["best_cr"] >= 0.5
    assert out["heldout"]["X_test"] is None and out["heldout"]["mask_val"].shape[0] == 16       # 48 utterances, every third
    assert out["test_conf"].sum() == 16                       # the confusion matrix kept is the validation one
    line = open(res_file).read().strip().split(",")
    assert len(line) == 2 and abs(float(line[0]) - out["best_cr"]) < 1e-9
    values = load_model(best_file)
    assert len(values) == len(net.params)
    net.close()


def test_extract_tools_round_trip(tmp_path):
    """runners/extract_encoder_from_model.py / extract_lstm_from_model.py: a saved 1-stream model -> the .mat files the
    stream loaders (load_decoder) and create_pretrained_model read."""
    from ip_avsr_amd.modelzoo import deltanet_majority_vote
    from ip_avsr_amd.runners import extract_encoder_from_model, extract_lstm_from_model
    from ip_avsr_amd.runners.nstream import load_decoder
    from ip_avsr_amd.utils.io import save_model_params
    rng = np.random.RandomState(2)
    dims = [24, 16, 12, 8, 5]
    ae = ([rng.normal(0, 0.3, (a, b)).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])],
          [rng.normal(0, 0.05, (b,)).astype(np.float32) for b in dims[1:]], dims[1:], ["rectify", "rectify", "rectify", "linear"])
    net = deltanet_majority_vote.create_model(ae, (None, None, 24), None, (None, None), None, 6, None, 4, 'glorot', False, True)
    pkl = str(tmp_path / "model.pkl")
    save_model_params(net, pkl)
    common = ["--shape", "16,12,8,5", "--input_dim", "24", "--lstm_size", "6", "--output_classes", "4", "--use_blstm"]
    enc_mat, lstm_mat = str(tmp_path / "enc.mat"), str(tmp_path / "lstm.mat")
    d = extract_encoder_from_model.main(common + ["--output", enc_mat, pkl])
    assert sorted(d) == ["b1", "b2", "b3", "b4", "w1", "w2", "w3", "w4"]
    w, b, shapes, _ = load_decoder(enc_mat, "16,12,8,5", "rectify,rectify,rectify,linear")
    for k in range(4):
        np.testing.assert_array_equal(w[k], ae[0][k]); np.testing.assert_array_equal(b[k], ae[1][k])
    lstm_names = [p.name.split(".")[0] for p in net.params if ".W_in_to_ingate" in p.name]
    l = extract_lstm_from_model.main(common + ["--layer_names", ",".join(lstm_names), "--output", lstm_mat, pkl])
    assert len(l) == 24 and "f_lstm_w_hid_to_cell" in l and "b_lstm_b_outgate" in l
    np.testing.assert_array_equal(l["f_lstm_w_in_to_ingate"], net.get_param(lstm_names[0] + ".W_in_to_ingate"))
    stored = sio.loadmat(lstm_mat)
    np.testing.assert_array_equal(stored["b_lstm_w_hid_to_forgetgate"], net.get_param(lstm_names[1] + ".W_hid_to_forgetgate"))
    net.close()


def test_variable_lr_runner_explodes_the_encoder_rates_after_epoch_4(tmp_path):
    """runners/1stream_variable_lr.py: fc1..fc3 train at their own rate; after epoch 4 the script sets them to 100.0 and
    the training cost must jump (that is the script's point)."""
    import importlib
    root = str(tmp_path)
    make_dataset(root, 1)
    ini = INI.format(k=1, root=root, reorder=True, diff=False) + TAIL.format(root=root, fusion="none", dropout=False)
    ini = ini.replace("num_epoch = 6", "num_epoch = 7").replace("validation_window = 4", "validation_window = 7")
    cfg = os.path.join(root, "cfg.ini")
    open(cfg, "w").write(ini)
    mod = importlib.import_module("ip_avsr_amd.runners.1stream_variable_lr")
    out = mod.run(["--config", cfg, "--seed", "3"])
    ct = out["cost_train"]
    assert len(ct) >= 6 and np.isfinite(ct[:4]).all()
    # once the rates are 100 the encoder is destroyed: the cost (bounded by log C for this loss) jumps back up
    assert ct[3] < ct[0] and (min(ct[4:]) > ct[3] + 0.1 or not np.isfinite(ct[4:]).all())
    out["network"].close()


@pytest.mark.parametrize("variant", ["noencoder", "dct"])
def test_one_stream_runner_variants(tmp_path, variant):
    """runners/1stream_noencoder.py (deltanet_v1: delta layer + BLSTM straight on the features) and 1stream_dct.py (host
    deltas before the split, lstm_classifier_majority_vote on 3x the features)."""
    from ip_avsr_amd.runners import nstream
    root = str(tmp_path)
    make_dataset(root, 1)
    ini = INI.format(k=1, root=root, reorder=False, diff=False) + TAIL.format(root=root, fusion="none", dropout=False)
    cfg = os.path.join(root, "cfg.ini")
    open(cfg, "w").write(ini)
    out = nstream.main(1, ["--config", cfg, "--seed", "5"], variant=variant)
    net = out["network"]
    names = [p.name for p in net.params]
    assert not any(n.startswith("fc") for n in names)                       # no encoder in either variant
    w = net.get_param("f_lstm.W_in_to_ingate")
    assert w.shape[0] == 72                                                   # 24 features x (static, delta, delta-delta)
    assert net.spec["streams"][0]["delta"] is (variant == "noencoder")        # device delta layer vs host deltas
    assert np.isfinite(out["cost_val"]).all() and (min(out["cost_val"]) < out["cost_val"][0] or out["best_cr"] >= 0.5)
    net.close()


@pytest.mark.parametrize("variant", ["dct", "nodelta"])
def test_two_stream_runner_variants(tmp_path, variant):
    """runners/2stream_dct.py (adenet_v2: stream 2 is an encoder-less feature stream, its section has no model /
    imagesize keys) and 2stream_nodelta.py (adenet_v2_nodelta: no delta layers)."""
    from ip_avsr_amd.runners import nstream
    root = str(tmp_path)
    make_dataset(root, 2)
    s1 = INI.format(k=1, root=root, reorder=True, diff=False)
    s2 = INI.format(k=2, root=root, reorder=False, diff=True)
    if variant == "dct":
        s2 = "\n".join(l for l in s2.split("\n") if not l.startswith(("model", "shape", "nonlinearities", "imagesize")))
    ini = s1 + s2 + TAIL.format(root=root, fusion="sum", dropout=False)
    cfg = os.path.join(root, "cfg.ini")
    open(cfg, "w").write(ini)
    out = nstream.main(2, ["--config", cfg, "--seed", "11"], variant=variant)
    net = out["network"]
    if variant == "dct":
        assert net.spec["streams"][1]["enc_shapes"] == [] and net.spec["streams"][1]["delta"] is True
        assert net.get_param("lstm_dct.W_in_to_ingate").shape[0] == 72
    else:
        assert all(s["delta"] is False for s in net.spec["streams"])
        assert net.get_param("lstm_s1.W_in_to_ingate").shape[0] == 5          # bottleneck features, no deltas
    assert np.isfinite(out["cost_val"]).all() and (min(out["cost_val"]) < out["cost_val"][0] or out["best_cr"] >= 0.5)
    net.close()


def test_the_drivers_default_arithmetic_is_bf16x3_on_the_weight_stationary_kernels(tmp_path, monkeypatch):
    """`python ip_avsr_amd/runners/3stream.py --config ...` WITHOUT --precision / ADN_PRECISION runs the fast
    fp32-grade arithmetic (bf16x3: modelzoo/_factory.PRODUCT_DEFAULT_PRECISION), i.e. the weight-stationary LSTM kernels -- not the
    exact-product diagnostic mode's one launch per time step (family counters of include/adenet.h adn_debug_lstm_family_counts:
    [0] per-step launches, [3] weight-stationary bf16x3)."""
    import ctypes as C
    from ip_avsr_amd import _lib
    from ip_avsr_amd.runners import nstream
    monkeypatch.delenv("ADN_PRECISION", raising=False)            # (tests/conftest.py asks for f32 for the exact-product tests)
    root = str(tmp_path)
    make_dataset(root, 3)
    ini = "".join(INI.format(k=k, root=root, reorder=(k == 1), diff=(k == 2)) for k in range(1, 4))
    ini += TAIL.format(root=root, fusion="concat", dropout=False)
    cfg = os.path.join(root, "cfg.ini")
    open(cfg, "w").write(ini)
    lib = _lib.load()

    def families():
        f, b = (C.c_int64 * 4)(), (C.c_int64 * 4)()
        lib.adn_debug_lstm_family_counts(f); lib.adn_debug_lstm_backward_family_counts(b)
        return np.array(list(f) + list(b))

    before = families()
    out = nstream.main(3, ["--config", cfg, "--seed", "7"])
    ran = families() - before
    assert out["network"].spec["precision"] == "bf16x3"
    out["network"].close()
    if not any(os.environ.get(k) for k in ("ADN_LSTM_NO_CLUSTER", "ADN_LSTM_NO_X3_CLUSTER", "ADN_LSTM_NO_X3_CLUSTER_BWD", "ADN_X3_NO_PLANES", "ADN_LSTM_CUS")):
        assert ran[3] > 0 and ran[7] > 0 and ran[0] == 0 and ran[4] == 0, ran
    assert np.isfinite(out["cost_val"]).all()
