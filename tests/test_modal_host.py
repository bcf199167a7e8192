"""Host logic of the schema-2 drivers (ip_avsr_amd/runners/modal.py): option precedence, the data paths of the four
scripts on synthetic files with the reference's .mat / .ini keys (SURVEY.md App. B schema 2, App. C), the update-rule
switch.  No device."""
import configparser

import numpy as np
import pytest

from ip_avsr_amd.runners import modal
from tests import modal_fixtures as MF


def _cfg(path, argv=()):
    options = modal.parse_options(list(argv) + ["--config", path], "unused.ini")
    config = configparser.ConfigParser()
    assert config.read(path)
    return modal._Cfg(config, options)


def test_cli_overrides_training_keys_only(tmp_path):
    _, bi = MF.make_avletters(str(tmp_path))
    cfg = _cfg(bi, ["--learning_rate", "0.5", "--update_rule", "adam", "--t1", "7"])
    assert cfg.get("training", "learning_rate", float) == 0.5 and cfg.get("training", "update_rule") == "adam"
    assert cfg.get("training", "t1", int) == 7 and cfg.get("training", "decay_rate", float) == 0.8
    assert cfg.get("models", "lstm_size", int) == 6
    assert cfg.get("training", "absent_key", int, 12) == 12
    with pytest.raises(configparser.NoOptionError):
        cfg.get("training", "absent_key", int)


def test_cuave_presplit_loader(tmp_path):
    """cuave/bimodal_with_val.py:211-249: +1 on the stored targets, reorder, per-sequence mean removal, per-frame
    z-normalisation; DCT features normalised with the TRAIN statistics."""
    cfg = _cfg(MF.make_cuave(str(tmp_path)))
    split, ys, lens = modal._load_cuave(cfg)
    assert [len(lens[k]) for k in ("train", "val", "test")] == [32, 12, 12]
    assert set(ys["train"]) == set(range(MF.CLASSES)) and ys["train"].shape == (int(lens["train"].sum()),)   # per frame
    for k in split:
        X, dct = split[k]
        assert X.shape == (int(lens[k].sum()), MF.D) and dct.shape == (int(lens[k].sum()), MF.DCT)
        assert np.allclose(X.mean(1), 0, atol=1e-5) and np.allclose(X.std(1), 1, atol=1e-3)
    assert np.allclose(split["train"][1].mean(0), 0, atol=1e-5) and np.allclose(split["train"][1].std(0), 1, atol=1e-3)
    assert abs(split["val"][1].mean()) > 1e-4                        # (train statistics, not its own)


def test_oulu_subject_split_loader(tmp_path):
    """oulu/trimodal_with_val.py:274-334: streams raw / dct / diff, subjects 1-5 / 6-7 / 8."""
    cfg = _cfg(MF.make_oulu(str(tmp_path)))
    split, ys, lens = modal._load_subject_split(cfg)
    assert [len(lens[k]) for k in ("train", "val", "test")] == [30, 12, 6]
    for k in split:
        raw, dct, diff = split[k]
        n = int(np.sum(lens[k]))
        assert raw.shape == (n, MF.D) and dct.shape == (n, MF.DCT) and diff.shape == (n, MF.D)
        assert np.allclose(raw.mean(1), 0, atol=1e-5)                # per-frame z-normalisation of the raw split only
    d = split["train"][2]
    assert np.array_equal(d[0], d[1])                                # frame 0 carries a copy of the first difference (:514)


def test_avletters_iter_split_loader(tmp_path):
    tri, bi = MF.make_avletters(str(tmp_path))
    split, ys, lens = modal._load_avletters(_cfg(tri), with_diff=True, normalise_images=False, target_offset=False)
    assert len(lens["train"]) == 32 and len(lens["test"]) == 16 and split["val"] is split["test"]
    assert len(split["train"]) == 3 and split["train"][2].shape == split["train"][0].shape
    split2, ys2, _ = modal._load_avletters(_cfg(bi), with_diff=False, normalise_images=True, target_offset=True)
    assert len(split2["train"]) == 2 and np.array_equal(ys2["train"], ys["train"])      # 1-based file, offset removed
    assert np.allclose(split2["train"][0].mean(1), 0, atol=1e-5)


class _FakeNet(object):
    def __init__(self):
        self.calls = []

    def train_step(self, ins, y, m, w, lr, want_loss=True):
        self.calls.append(("adam", lr)); return 1.0 if want_loss else None

    def compute_grads(self, ins, y, m, w, want_loss=True):
        self.calls.append(("grads",)); return 2.0 if want_loss else None

    def apply_adadelta(self, lr):
        self.calls.append(("adadelta", lr))

    def apply_sgd(self, lr, mm, nesterov=False):
        self.calls.append(("sgd", lr, mm, nesterov))


def test_update_rule_switch():
    """avletters/bimodal.py:446-455: adadelta(lr) | sgd + momentum | sgd + nesterov momentum | adam with DEFAULT parameters."""
    for rule, want in (("adadelta", ("adadelta", 0.3)), ("sgdm", ("sgd", 0.3, 0.6, False)), ("sgdnm", ("sgd", 0.3, 0.6, True)),
                       ("adam", ("adam", 1e-3))):
        net = _FakeNet()
        up = modal.Updater(net, rule, 0.3, 0.6)
        assert up(None, None, None, 9) is None            # the loops discard train's cost: not waited for by default
        assert net.calls[-1] == want, rule
        assert up(None, None, None, 9, want_loss=True) in (1.0, 2.0)
    up.lr = 0.1
    with pytest.raises(ValueError):
        modal.Updater(_FakeNet(), "rmsprop", 0.1)


def test_load_ae_accepts_mat_and_pickle(tmp_path):
    rng = np.random.RandomState(0)
    a = modal.load_ae(MF._ae(rng, str(tmp_path), "a.mat"))
    b = modal.load_ae(MF._ae(rng, str(tmp_path), "b.pkl", as_pickle=True))
    for w, bias in (a, b):
        assert [x.shape for x in w] == [(24, 16), (16, 12), (12, 8), (8, 5)] and [x.shape for x in bias] == [(16,), (12,), (8,), (5,)]


# ----------------------------------------------------------------------------------------------- round 3: the rest of the family
def test_every_reference_script_of_the_family_has_a_plan_and_an_entry_point():
    import importlib
    assert set(modal.SCRIPTS) == set(modal._PLANS)
    for ds, sc in modal.SCRIPTS:
        mod = importlib.import_module("ip_avsr_amd.%s.%s" % (ds, sc))
        assert callable(mod.main)
    assert modal.SCRIPTS[("cuave", "audio_visual_runner")] == "config/avnet.ini"
    assert modal.SCRIPTS[("avletters", "unimodal")] == "config/normal.ini"
    with pytest.raises(ValueError):
        modal.main("cuave", "no_such_script", [])


def test_key_aliases_of_the_scripts(tmp_path):
    uni, bi = MF.make_oulu_family(str(tmp_path))
    cfg = _cfg(uni, ["--no_epochs", "9"])
    assert cfg.get("training", ("num_epoch", "no_epochs"), int) == 9                    # CLI wins, under either spelling
    assert cfg.get("models", ("lstm_size", "lstm_units"), int) == 8
    cfg = _cfg(bi)
    assert cfg.get("models", ("no_coeffs", "no_coeff"), int) == MF.DCT and cfg.get("training", ("num_epoch", "no_epochs"), int) == 3


def test_cuave_subject_unimodal_loader(tmp_path):
    """cuave/unimodal_with_val.py:201-244: subjects 1-5 / 6-7 / 8, targets + 1, per-frame z-normalised, per-sequence mean
    removed (so every utterance's frames sum to ~0 BEFORE the z-normalisation; after it each frame has zero mean)."""
    cfg = _cfg(MF.make_cuave_subject(str(tmp_path)))
    split, ys, lens = modal._load_cuave_subject_unimodal(cfg)
    assert [len(lens[k]) for k in ("train", "val", "test")] == [30, 12, 6]
    assert set(np.unique(ys["train"])) == set(range(MF.CLASSES))
    for k in split:
        (X,) = split[k]
        assert X.shape == (int(np.sum(lens[k])), MF.D) and np.allclose(X.mean(1), 0, atol=1e-5) and np.allclose(X.std(1), 1, atol=1e-3)


def test_cuave_family_loaders(tmp_path):
    dct, tri, av = MF.make_cuave_family(str(tmp_path))
    split, ys, lens = modal._load_cuave_dct(_cfg(dct))
    assert len(split["train"]) == 1 and split["train"][0].shape == (int(lens["train"].sum()), MF.DCT)
    assert np.allclose(split["train"][0].mean(0), 0, atol=1e-5) and set(ys["val"]) == set(range(MF.CLASSES))
    split, ys, lens = modal._load_cuave_trimodal(_cfg(tri))
    raw, d, diff = split["test"]
    assert raw.shape == diff.shape and d.shape[1] == MF.DCT and np.array_equal(diff[0], diff[1])
    assert abs(raw.mean(1)).max() > 1e-3                                                # raw frames as stored: NOT normalised
    l0 = int(lens["test"][0])
    assert np.allclose(diff[1:l0], raw[1:l0] - raw[:l0 - 1])
    split, ys, lens = modal._load_cuave_av(_cfg(av))
    vis, aud = split["train"]
    assert vis.shape == (int(lens["train"].sum()), MF.D) and aud.shape == (int(lens["train"].sum()), 14)
    stored = MF.sio.loadmat(str(tmp_path / "cuave.mat"))["trData"].astype("float32")
    assert np.array_equal(vis, stored.reshape(-1, 6, 4).transpose(0, 2, 1).reshape(-1, 24))   # F -> C pixel order, nothing else


def test_oulu_and_avletters_family_loaders(tmp_path):
    uni, bi = MF.make_oulu_family(str(tmp_path))
    split, ys, lens = modal._load_oulu(_cfg(uni), with_dct=False)
    assert [len(lens[k]) for k in ("train", "val", "test")] == [30, 12, 6] and len(split["val"]) == 1
    split, ys, lens = modal._load_oulu(_cfg(bi), with_dct=True)
    assert np.allclose(split["train"][0].mean(1), 0, atol=1e-5) and np.allclose(split["train"][1].mean(0), 0, atol=1e-5)
    diff, enc, raw = MF.make_avletters_family(str(tmp_path))
    split, ys, lens = modal._load_avletters_diff(_cfg(diff))
    assert len(split["train"]) == 2 and len(lens["train"]) == 32 and len(lens["test"]) == 16 and split["val"] is split["test"]
    config = configparser.ConfigParser(); config.read(raw)
    split, ys, lens = modal._load_avletters_stream1(config)
    assert set(np.unique(ys["train"])) == set(range(MF.CLASSES))                        # matlab_target_offset removed the 1
    assert np.allclose(split["train"][0].mean(0), 0, atol=1e-4)                         # featurewisenormalize, train statistics
    config = configparser.ConfigParser(); config.read(enc)
    split2, _, _ = modal._load_avletters_stream1(config)
    assert np.allclose(split2["train"][0].mean(1), 0, atol=1e-5)                        # samplewisenormalize only
