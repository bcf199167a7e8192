"""End-to-end runs of the schema-2 drivers (reference cuave/bimodal_with_val.py, oulu/trimodal_with_val.py,
avletters/trimodal.py, avletters/bimodal.py) on an MI355X with synthetic files in the reference's .mat / .ini schema."""
import os

import numpy as np
import pytest

from tests import modal_fixtures as MF

pytestmark = pytest.mark.gpu


def test_cuave_bimodal_with_val(tmp_path):
    from ip_avsr_amd.cuave import bimodal_with_val
    root = str(tmp_path)
    res = os.path.join(root, "res.csv")
    out = bimodal_with_val.main(["--config", MF.make_cuave(root), "--write_results", res, "--seed", "3", "--no_plot"])
    net = out["network"]
    assert net.head == "frames" and net.S == 2 and net.spec["fusion"] == "adasum"      # adenet_v2: encoder + DCT stream
    assert "lstm_bn.W_cell_to_ingate" in net.param_index                                # use_peepholes = True
    assert len(out["cost_val"]) >= 2 and np.isfinite(out["cost_val"]).all() and out["test_cr"] is not None
    assert min(out["cost_val"]) < out["cost_val"][0] or out["best_cr"] >= 0.5
    lines = open(res).read().strip().split("\n")
    head = lines[0].split(",")
    assert len(head) == 12 and head[3] == "adam" and head[5] == "RELU" and abs(float(head[10]) - 100 * out["best_cr"]) < 1e-9
    assert len(lines) == 4 and len(lines[1].split(",")) == len(out["cost_train"])
    net.close()


def test_oulu_trimodal_with_val(tmp_path):
    from ip_avsr_amd.oulu import trimodal_with_val
    out = trimodal_with_val.main(["--config", MF.make_oulu(str(tmp_path)), "--seed", "5", "--no_plot"])
    net = out["network"]
    assert net.head == "last" and net.S == 3 and net.H == 12                            # adenet_v3: lstm_size / (1 - 0.5)
    assert "lstm_raw.W_cell_to_outgate" in net.param_index                              # Lasagne's default peepholes
    assert len(out["cost_val"]) >= 2 and np.isfinite(out["cost_val"]).all()
    n = len(out["cost_val"])
    # adadelta's learning rate: 1.0, multiplied by 0.5 after every epoch from decay_start = 2 on (:508-510)
    assert abs(out["learning_rate"] - 0.5 ** max(0, n - 1)) < 1e-6
    net.close()


def test_avletters_trimodal_and_bimodal(tmp_path):
    from ip_avsr_amd.avletters import bimodal, trimodal
    tri, bi = MF.make_avletters(str(tmp_path))
    out = trimodal.main(["--config", tri, "--seed", "1", "--no_plot"])
    assert out["network"].head == "last" and out["test_cr"] is None and np.isfinite(out["cost_val"]).all()
    out["network"].close()
    res = os.path.join(str(tmp_path), "bi.csv")
    out = bimodal.main(["--config", bi, "--seed", "1", "--no_plot", "--write_results", res])
    assert out["network"].head == "frames" and np.isfinite(out["cost_val"]).all()
    assert out["momentum"] in (0.5, 0.7, 0.9) and out["learning_rate"] <= 0.05          # sgdm schedule fired or not, never above
    assert len(open(res).read().strip().split("\n")) == 5
    out["network"].close()
    out = bimodal.main(["--config", bi, "--seed", "1", "--no_plot", "--update_rule", "adadelta", "--learning_rate", "1.0"])
    assert np.isfinite(out["cost_val"]).all()
    out["network"].close()


# ----------------------------------------------------------------------------------------------- round 3: the rest of the family
def test_cuave_unimodal_scripts(tmp_path):
    """cuave/unimodal_with_val.py (encoder + BLSTM, adam(lr), --save_best) and cuave/unimodal_dct_with_val.py (BLSTM on the DCT
    features, adam with default parameters)."""
    from ip_avsr_amd.cuave import unimodal_dct_with_val, unimodal_with_val
    from ip_avsr_amd.utils.io import load_model_params
    root = str(tmp_path)
    res, best = os.path.join(root, "uni.csv"), os.path.join(root, "best.pkl")
    out = unimodal_with_val.main(["--config", MF.make_cuave_subject(root), "--write_results", res, "--save_best", best, "--seed", "3",
                                  "--no_plot"])
    net = out["network"]
    assert net.head == "frames" and net.S == 1 and "f_blstm1.W_in_to_ingate" in net.param_index
    assert np.isfinite(out["cost_val"]).all() and out["test_cr"] is not None
    line = open(res).read().strip().split(",")
    assert len(line) == 3 and float(line[0]) == out["test_cr"] and float(line[2]) == out["best_val"]
    saved = net.get_all_param_values()                                  # --save_best restored the best snapshot and pickled it
    load_model_params(net, best)
    for a, b in zip(saved, net.get_all_param_values()):
        np.testing.assert_array_equal(a, b)
    net.close()
    dct, _, _ = MF.make_cuave_family(root)
    out = unimodal_dct_with_val.main(["--config", dct, "--write_results", res + "2", "--seed", "4", "--no_plot"])
    net = out["network"]
    assert net.S == 1 and not net.spec["streams"][0]["delta"] and not net.spec["streams"][0]["enc_names"]
    assert "f_lstm.W_hid_to_cell" in net.param_index and np.isfinite(out["cost_val"]).all()
    lines = open(res + "2").read().strip().split("\n")
    assert len(lines) == 4 and lines[0].split(",")[5] == "N/A" and lines[0].split(",")[3] == "adam"
    net.close()


def test_cuave_trimodal_and_audio_visual_runner(tmp_path):
    from ip_avsr_amd.cuave import audio_visual_runner, trimodal_with_val
    root = str(tmp_path)
    _, tri, av = MF.make_cuave_family(root)
    res = os.path.join(root, "tri.csv")
    out = trimodal_with_val.main(["--config", tri, "--write_results", res, "--seed", "2", "--no_plot"])
    net = out["network"]
    assert net.head == "last" and net.S == 3 and net.spec["fusion"] == "adasum" and net.H == 12
    n = len(out["cost_val"])
    assert np.isfinite(out["cost_val"]).all() and abs(out["learning_rate"] - 0.5 ** max(0, n - 1)) < 1e-6
    line = open(res).read().strip().split(",")
    assert line[0] == "adasum" and float(line[1]) == out["test_cr"]
    net.close()
    res = os.path.join(root, "av.csv")
    out = audio_visual_runner.main(["--config", av, "--write_results", res, "--seed", "2", "--no_plot"])
    net = out["network"]
    assert net.head == "frames" and net.S == 2 and net.spec["fusion"] == "concat"
    assert [s["input_dim"] for s in net.spec["streams"]] == [MF.D, 14]
    assert "lstm_visual.W_cell_to_ingate" in net.param_index and "f_lstm_agg.W_cell_to_ingate" not in net.param_index
    assert "bottleneck_audio.W" in net.param_index and np.isfinite(out["cost_val"]).all()
    lines = open(res).read().strip().split("\n")
    assert len(lines) == 4 and lines[0].split(",")[5] == "RELU" and len(lines[0].split(",")) == 12
    assert min(out["cost_val"]) < out["cost_val"][0] or out["best_cr"] >= 0.5
    net.close()


def test_oulu_unimodal_and_bimodal_with_val(tmp_path):
    from ip_avsr_amd.oulu import bimodal_with_val, unimodal_with_val
    uni, bi = MF.make_oulu_family(str(tmp_path))
    out = unimodal_with_val.main(["--config", uni, "--seed", "1", "--no_plot"])
    net = out["network"]
    assert net.S == 1 and net.H == 8 and "f_blstm1.W_cell_to_outgate" in net.param_index       # use_peepholes = True
    assert np.isfinite(out["cost_val"]).all() and len(out["cost_val"]) <= 4
    net.close()
    res = os.path.join(str(tmp_path), "bi.csv")
    out = bimodal_with_val.main(["--config", bi, "--seed", "1", "--no_plot", "--write_results", res])
    net = out["network"]
    assert net.S == 2 and net.spec["fusion"] == "sum" and net.spec["streams"][1]["input_dim"] == MF.DCT
    assert np.isfinite(out["cost_val"]).all() and open(res).read().startswith("sum,")
    net.close()


def test_avletters_bimodal_diff_image_and_unimodal(tmp_path):
    from ip_avsr_amd.avletters import bimodal_diff_image, unimodal
    diff, enc, raw = MF.make_avletters_family(str(tmp_path))
    res = os.path.join(str(tmp_path), "diff.csv")
    out = bimodal_diff_image.main(["--config", diff, "--seed", "1", "--no_plot", "--write_results", res])
    net = out["network"]
    assert net.head == "last" and net.S == 2 and net.spec["fusion"] == "adasum"
    assert "lstm_diff.W_cell_to_ingate" in net.param_index                      # the script's CLI default forces peepholes on
    assert np.isfinite(out["cost_val"]).all() and out["test_cr"] is None
    n = len(out["cost_val"])
    assert out["learning_rate"] <= 0.05 * 0.8 ** max(0, n - 1) + 1e-6           # decay from decay_start = 2 (+ the t1 rule)
    lines = open(res).read().strip().split("\n")
    assert len(lines) == 5 and lines[0].split(",")[0] == "sgdnm" and lines[-1].startswith("adasum,")
    net.close()
    out = bimodal_diff_image.main(["--config", diff, "--seed", "1", "--no_plot", "--update_rule", "adam"])
    assert np.isfinite(out["cost_val"]).all()
    out["network"].close()
    out = unimodal.main(["--config", enc, "--seed", "1", "--no_plot"])
    net = out["network"]
    assert net.S == 1 and net.spec["streams"][0]["enc_names"] == ["fc1", "fc2", "fc3", "bottleneck"] and out["test_cr"] is None
    assert "f_blstm1.b_cell" in net.param_index and np.isfinite(out["cost_val"]).all()
    net.close()
    out = unimodal.main(["--config", raw, "--seed", "1", "--no_plot"])
    net = out["network"]
    assert not net.spec["streams"][0]["enc_names"] and net.spec["streams"][0]["delta"] and "lstm.W_in_to_cell" in net.param_index
    net.close()


def test_avnet_matches_the_oracle(tmp_path):
    """create_pretrained_substream x 2 -> create_model (modelzoo/avnet.py:30-114) against the oracle's N-stream graph with
    the same layer names: probabilities, loss and every gradient, all three fusion types, with and without peepholes."""
    from oracle import adenet_oracle as O
    from ip_avsr_amd.modelzoo import avnet
    rng = np.random.default_rng(7)
    Dv, Da, H, C, B, T, theta = 24, 14, 10, 4, 5, 9, 2
    shapes = (16, 12, 8, 5)
    lens = rng.integers(3, T + 1, size=B); lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    xs = [(rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32) for d in (Dv, Da)]
    y = np.repeat(rng.integers(0, C, size=(B, 1)), T, axis=1).astype(np.int32)
    for fusion, peep in (("concat", True), ("adasum", False), ("sum", True)):
        spec = O.spec_nstream([Dv, Da], enc_shapes=shapes, enc_acts=("sigmoid", "sigmoid", "sigmoid", "linear"), lstm_size=H,
                              classes=C, fusion=fusion, peepholes=peep)
        for s, name in zip(spec["streams"], ("visual", "audio")):           # avnet's layer names
            s["enc_names"] = ["%s_%s" % (n, name) for n in ("fc1", "fc2", "fc3", "bottleneck")]
            s["lstm_names"] = ["lstm_" + name]
        p = O.init_params(spec, rng, np.float64, enc_std=0.3, perturb=0.1)
        subs = []
        for s, d, name in zip(spec["streams"], (Dv, Da), ("visual", "audio")):
            ws = [p[n + ".W"].astype(np.float32) for n in s["enc_names"]]
            bs = [p[n + ".b"].astype(np.float32) for n in s["enc_names"]]
            subs.append(avnet.create_pretrained_substream(ws, bs, (None, None, d), None, (None, None), None, name, H, None,
                                                          "sigmoid", "glorot", peep))
        net, l_fuse = avnet.create_model(subs, (None, None), None, H, C, fusion, "glorot", peep)
        assert [q.name for q in net.params] == O.param_names(spec)
        net.set_params_dict({k: v.astype(np.float32) for k, v in p.items()})
        x64 = [x.astype(np.float64) for x in xs]
        ref = O.forward(spec, p, x64, mask, theta)
        np.testing.assert_allclose(net.predict(xs, mask, theta), ref, atol=2e-5)
        l_ref, g_ref, _ = O.loss_and_grads(spec, p, x64, y, mask, theta)
        l = net.compute_grads(xs, y, mask, theta)
        assert abs(l - l_ref) < 1e-5 * abs(l_ref)
        g = net.get_grads_dict()
        for k in g_ref:
            scale = max(1e-6, np.abs(g_ref[k]).max())
            assert np.abs(g[k] - g_ref[k]).max() <= 1e-4 * scale + 1e-7, (fusion, k)
        if fusion == "adasum":
            assert np.allclose(np.ravel(l_fuse.get_all_param_values(scaling_param=True)), [p["adasum1.adacoeff0"], p["adasum1.adacoeff1"]])
        net.close()


# ------------------------------------------------------------------- the drivers' training follows the oracle's update rule
def _run_and_replay(monkeypatch, run, tol):
    """Runs a schema-2 driver with a spy on the network's training calls (records the parameters ahead of the first step and
    every minibatch, rule, learning rate and momentum the driver hands over; pins the dropout stream to (seed 4321, call number)),
    then replays the SAME batches through the oracle's graph and update rules in float64 from the same parameters and
    compares every trained tensor."""
    from oracle import adenet_oracle as O
    from ip_avsr_amd.model import AdeNetModel
    rec = dict(calls=[], p0=None, spec=None)
    plain = {k: getattr(AdeNetModel, k) for k in ("train_step", "compute_grads", "apply_sgd", "apply_adadelta")}

    def note(net, inputs, targets, mask, window, **rule):
        if rec["p0"] is None:
            rec["p0"] = {k: np.asarray(v, np.float64) for k, v in net.get_params_dict().items()}
            rec["spec"] = net.spec
        net.set_dropout_state(4321, len(rec["calls"]))
        host = lambda x: x.float().cpu().numpy() if hasattr(x, "cpu") else x      # (the drivers hand over HBM-resident batches)
        rec["calls"].append(dict(xs=[np.array(host(x), np.float64) for x in inputs], y=np.array(targets), mask=np.array(mask),
                                 window=int(window), **rule))

    def train_step(self, inputs, targets, mask, window, learning_rate, *a, **k):      # forward + backward + Adam
        note(self, inputs, targets, mask, window, rule="adam", lr=float(learning_rate), mm=0.0)
        return plain["train_step"](self, inputs, targets, mask, window, learning_rate, *a, **k)

    def compute_grads(self, inputs, targets, mask, window, *a, **k):                  # ... the other rules: gradients,
        note(self, inputs, targets, mask, window, rule=None)
        return plain["compute_grads"](self, inputs, targets, mask, window, *a, **k)

    def apply_sgd(self, learning_rate, momentum=0.0, nesterov=False):                 # ... then one of these
        rec["calls"][-1].update(rule="sgdnm" if nesterov else "sgdm", lr=float(learning_rate), mm=float(momentum))
        return plain["apply_sgd"](self, learning_rate, momentum, nesterov)

    def apply_adadelta(self, learning_rate=1.0, *a, **k):
        rec["calls"][-1].update(rule="adadelta", lr=float(learning_rate), mm=0.0)
        return plain["apply_adadelta"](self, learning_rate, *a, **k)

    for k, f in (("train_step", train_step), ("compute_grads", compute_grads), ("apply_sgd", apply_sgd), ("apply_adadelta", apply_adadelta)):
        monkeypatch.setattr(AdeNetModel, k, f)
    out = run()
    for k, f in plain.items():
        monkeypatch.setattr(AdeNetModel, k, f)
    net, spec, p = out["network"], rec["spec"], {k: v.copy() for k, v in rec["p0"].items()}
    assert list(p) == O.param_names(spec)
    has_dropout = bool(spec.get("agg_dropout")) or any(s.get("dropout") for s in spec["streams"])
    adam, vel, ada = O.adam_init(p), O.momentum_init(p), O.adadelta_init(p)
    for k, c in enumerate(rec["calls"]):
        _, g, cache = O.loss_and_grads(spec, p, c["xs"], c["y"], c["mask"], c["window"],
                                       dropout=dict(seed=4321, counter=k) if has_dropout else None, training=True)
        if c["rule"] == "adam":
            O.adam_step(p, g, adam, c["lr"])
        elif c["rule"] == "adadelta":
            O.adadelta_step(p, g, ada, c["lr"])
        else:
            O.momentum_step(p, g, vel, c["lr"], c["mm"], nesterov=(c["rule"] == "sgdnm"))
        O.bn_running_update(spec, p, cache)
    got = net.get_params_dict()
    worst = 0.0
    for name in p:
        moved = np.abs(p[name] - rec["p0"][name]).max()
        err = np.abs(got[name] - p[name]).max()
        scale = max(np.abs(p[name]).max(), 1e-3)
        worst = max(worst, err / scale)
        assert err <= tol * scale, (name, err, scale, moved)
    trained = sum(np.abs(p[n] - rec["p0"][n]).max() > 1e-5 for n in p)
    assert trained >= len(p) - 2 * len(spec["streams"]) - 2, "the replay must have moved the parameters"   # (BatchNorm statistics aside)
    return out, rec, worst


def test_schema2_training_follows_the_oracle_update_rules(tmp_path, monkeypatch):
    """not only finite costs -- the parameters a schema-2 driver leaves behind are the ones the oracle's
    graph + update rule produce from the same initial parameters over the same minibatches, learning-rate / momentum
    schedules included: Adam at Lasagne's default rate (cuave/bimodal_with_val.py:333), sgd with classic momentum and the
    t1 / decay schedule, Nesterov momentum and adadelta (avletters/bimodal.py:446-455,541-545), adadelta with the per-epoch
    decay on the dropout model adenet_v3 (oulu/trimodal_with_val.py:388,508-510)."""
    from ip_avsr_amd.avletters import bimodal
    from ip_avsr_amd.cuave import bimodal_with_val
    from ip_avsr_amd.oulu import trimodal_with_val
    root = str(tmp_path)
    for sub in "cao":
        os.makedirs(os.path.join(root, sub))
    out, rec, worst = _run_and_replay(monkeypatch, lambda: bimodal_with_val.main(
        ["--config", MF.make_cuave(os.path.join(root, "c")), "--seed", "3", "--no_plot", "--no_epochs", "3"]), 2e-4)
    assert {c["rule"] for c in rec["calls"]} == {"adam"} and len(rec["calls"]) >= 3
    out["network"].close()
    tri, bi = MF.make_avletters(os.path.join(root, "a"))
    for extra in (["--update_rule", "sgdm"], ["--update_rule", "sgdnm"], ["--update_rule", "adadelta", "--learning_rate", "1.0"]):
        out, rec, worst = _run_and_replay(monkeypatch, lambda: bimodal.main(["--config", bi, "--seed", "1", "--no_plot"] + extra), 2e-4)
        assert {c["rule"] for c in rec["calls"]} == {extra[1]}
        if extra[1] == "adadelta":
            # avletters/bimodal.py:552-555 decays the rate of EVERY update rule from decay_start (= 3) on, behind the t1 rule
            # (which touches sgdm / sgdnm only): 1.0 -> 0.8 -> 0.64 ... (the driver had dropped this decay)
            n = len(out["cost_val"])
            assert n >= 4 and any(abs(out["learning_rate"] - 0.8 ** k) < 1e-5 for k in (n - 2, n - 3)), (n, out["learning_rate"])
            assert {round(c["lr"], 4) for c in rec["calls"]} >= {1.0, 0.8}
        out["network"].close()
    out, rec, worst = _run_and_replay(monkeypatch, lambda: trimodal_with_val.main(
        ["--config", MF.make_oulu(os.path.join(root, "o")), "--seed", "5", "--no_plot"]), 5e-4)
    assert any(s.get("dropout") for s in rec["spec"]["streams"]) or rec["spec"].get("agg_dropout")
    assert len({c["lr"] for c in rec["calls"]}) >= 2, "the per-epoch decay must have fired inside the recorded run"
    out["network"].close()
