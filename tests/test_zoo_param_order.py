"""Parameter lists of the zoo modules whose reference graphs rely on Lasagne's default ``peepholes=True``
(reference modelzoo/adenet_v3.py:27-45,113-143 and the same pattern in adenet_v4/v5/v6, baseline_end2end,
lstm_classifier_baseline: no ``peepholes=`` argument on any LSTMLayer).  A checkpoint written by the reference for one
of these models has W_cell_to_{ingate,forgetgate,outgate} after the twelve gate tensors of every LSTM;
``set_all_param_values`` only works if this package lists the same tensors in the same order.  Host only."""
import numpy as np
import pytest

from ip_avsr_amd.modelzoo import _factory as F


class _Layer(object):
    def __init__(self, W, b):
        self.W, self.b = W, b


class _Net(object):                                     # what nolearn hands over: .get_all_layers()[1..4].W / .b
    def __init__(self, d):
        dims = [d, 20, 12, 8, 5]
        self.layers = [None] + [_Layer(np.zeros((a, b), np.float32), np.zeros(b, np.float32))
                                for a, b in zip(dims[:-1], dims[1:])]

    def get_all_layers(self):
        return self.layers


@pytest.fixture
def spec_only(monkeypatch):
    monkeypatch.setattr(F, "SPEC_ONLY", True)


def lstm(name, peep=True):
    out = []
    for g in ("ingate", "forgetgate", "cell", "outgate"):
        out += ["%s.W_in_to_%s" % (name, g), "%s.W_hid_to_%s" % (name, g), "%s.b_%s" % (name, g)]
    if peep:
        out += [name + ".W_cell_to_ingate", name + ".W_cell_to_forgetgate", name + ".W_cell_to_outgate"]
    return out + [name + ".cell_init", name + ".hid_init"]


def enc(sfx):
    return [n + sfx + k for n in ("fc1", "fc2", "fc3", "bottleneck") for k in (".W", ".b")]


def test_adenet_v3_lists_the_peephole_vectors_in_lasagne_order(spec_only):
    from ip_avsr_amd.modelzoo import adenet_v3
    spec, _ = adenet_v3.create_model(_Net(30), _Net(30), (None, None, 30), None, (None, None), None, (None, None, 9), None,
                                     (None, None, 30), None, lstm_size=4, output_classes=5, fusiontype='adasum')
    want = (enc("_raw") + lstm("lstm_raw") + lstm("lstm_dct") + enc("_diff") + lstm("lstm_diff") +
            ["adasum1.adacoeff0", "adasum1.adacoeff1", "adasum1.adacoeff2"] + lstm("f_lstm_agg") + lstm("b_lstm_agg") +
            ["output.W", "output.b"])
    assert F.param_names(spec) == want
    assert len(want) == 2 * 8 + 5 * 17 + 3 + 2


@pytest.mark.parametrize("module,args,lstms", [
    ("adenet_v4", lambda: (_Net(30), (None, None, 30), None, (None, None), None, (None, None, 9), None),
     ["lstm_bn", "lstm_dct", "lstm_agg"]),
    ("adenet_v5", lambda: (_Net(30), _Net(30), (None, None, 30), None, (None, None), None, (None, None, 9), None,
                           (None, None, 30), None), ["lstm_raw", "lstm_dct", "lstm_diff", "f_lstm_agg", "b_lstm_agg"]),
    ("adenet_v6", lambda: (_Net(30), _Net(30), (None, None, 30), None, (None, None), None, (None, None, 30), None),
     ["lstm_raw", "lstm_diff", "f_lstm_agg", "b_lstm_agg"]),
    ("baseline_end2end", lambda: (_Net(30), (None, None, 30), None, (None, None), None), ["f_lstm1", "b_lstm1"]),
    ("lstm_classifier_baseline", lambda: ((None, None, 9), None, (None, None), None), ["f_lstm", "b_lstm"]),
])
def test_default_peepholes_of_the_inline_lstm_layers(spec_only, module, args, lstms):
    import importlib
    mod = importlib.import_module("ip_avsr_amd.modelzoo." + module)
    res = mod.create_model(*args())
    spec = res[0] if isinstance(res, tuple) else res
    names = F.param_names(spec)
    for ln in lstms:
        block = lstm(ln)
        i = names.index(block[0])
        assert names[i:i + len(block)] == block, ln
    assert sum(n.endswith(".W_cell_to_ingate") for n in names) == len(lstms)


def test_explicit_peepholes_false_is_still_honoured(spec_only):
    """modelzoo/adenet_v2.py:45-63 passes ``peepholes=use_peepholes`` (default False) explicitly."""
    from ip_avsr_amd.modelzoo import adenet_v2
    ae = ([np.zeros((30, 20), np.float32), np.zeros((20, 5), np.float32)], [np.zeros(20, np.float32), np.zeros(5, np.float32)],
          [20, 5], ["rectify", "linear"])
    res = adenet_v2.create_model(ae, (None, None, 30), None, (None, None), None, (None, None, 9), None, 4, None, 3)
    spec = res[0] if isinstance(res, tuple) else res
    assert not any("W_cell_to" in n for n in F.param_names(spec))


def test_adenet_v1_and_v1_1_parameter_lists(spec_only):
    """modelzoo/adenet_v1.py:48-109: fc1 .. bottleneck, batchnorm1 (beta, gamma, mean, inv_std), the two BLSTMs with
    Lasagne's default peepholes, 'output'.  v1: lstm_size units under 2 * lstm_size; v1_1: both 2 * lstm_size, dropout."""
    from ip_avsr_amd.modelzoo import adenet_v1, adenet_v1_1
    args = lambda: (_Net(30), (None, None, 30), None, (None, None), None, (None, None, 9), None)
    spec, _ = adenet_v1.create_model(*args(), lstm_size=4, output_classes=3)
    enc_names = [n + k for n in ("fc1", "fc2", "fc3", "bottleneck") for k in (".W", ".b")]
    want = (enc_names + ["batchnorm1.beta", "batchnorm1.gamma", "batchnorm1.mean", "batchnorm1.inv_std"] +
            lstm("f_lstm1") + lstm("b_lstm1") + lstm("f_lstm2") + lstm("b_lstm2") + ["output.W", "output.b"])
    assert F.param_names(spec) == want
    assert spec["lstm_size"] == 8 and spec["stream_lstm_size"] == 4 and spec["streams"][0]["aux_dim"] == 9
    spec11 = adenet_v1_1.create_model(*args(), lstm_size=4, output_classes=3)
    assert F.param_names(spec11) == want and spec11["stream_lstm_size"] == 8
    assert spec11["streams"][0]["dropout"] == 0.5 and spec11["agg_dropout"] == 0.5


def test_avnet_substreams_parameter_list_and_activations(spec_only):
    """modelzoo/avnet.py:30-114: ``create_pretrained_substream`` names its layers fc1_<name> .. bottleneck_<name> /
    lstm_<name> (:45-48,58-66), uses [nonlinearity] * 3 + [linear] (:47), passes ``peepholes=use_peepholes`` to the stream
    LSTM; ``create_model`` fuses the substreams in list order, its BLSTM comes from create_blstm WITHOUT the peephole
    argument (custom/layers.py:55: default False), the classifier is 'softmax'.  get_all_params order = depth first
    through the fusion layer's inputs."""
    from ip_avsr_amd.modelzoo import avnet
    w = lambda d: ([np.zeros((a, b), np.float32) for a, b in zip([d, 20, 12, 8], [20, 12, 8, 5])],
                   [np.zeros(b, np.float32) for b in (20, 12, 8, 5)])
    vw, vb = w(30)
    aw, ab = w(14)
    for fusion, peep in (("adasum", True), ("concat", False), ("sum", True)):
        vis = avnet.create_pretrained_substream(vw, vb, (None, None, 30), None, (None, None), None, 'visual', 6, None, 'sigmoid',
                                                'glorot', peep)
        aud = avnet.create_pretrained_substream(aw, ab, (None, None, 14), None, (None, None), None, 'audio', 6, None, 'sigmoid',
                                                'glorot', peep)
        assert vis["enc_acts"] == ["sigmoid", "sigmoid", "sigmoid", "linear"] and aud["input_dim"] == 14 and vis["delta"]
        spec, _ = avnet.create_model([vis, aud], (None, None), None, 6, 3, fusion, 'glorot', peep)
        want = (enc("_visual") + lstm("lstm_visual", peep) + enc("_audio") + lstm("lstm_audio", peep) +
                (["adasum1.adacoeff0", "adasum1.adacoeff1"] if fusion == "adasum" else []) +
                lstm("f_lstm_agg", False) + lstm("b_lstm_agg", False) + ["softmax.W", "softmax.b"])
        assert F.param_names(spec) == want
        assert spec["fusion"] == fusion and spec["fuse_name"] == {"adasum": "adasum1", "concat": "concat", "sum": "sum1"}[fusion]
        assert spec["lstm_size"] == 6 and spec["classes"] == 3 and spec.get("head", "frames") == "frames"


def test_unimodal_scripts_encoder_factory(spec_only):
    """``deltanet_majority_vote.create_model_using_pretrained_encoder`` as cuave/unimodal_with_val.py:259-263 calls it."""
    from ip_avsr_amd.modelzoo import deltanet_majority_vote as dmv
    ws = [np.zeros((a, b), np.float32) for a, b in zip([30, 20, 12, 8], [20, 12, 8, 5])]
    bs = [np.zeros(b, np.float32) for b in (20, 12, 8, 5)]
    spec = dmv.create_model_using_pretrained_encoder(ws, bs, (None, None, 30), None, (None, None), None, 6, None, 4, 'glorot', True,
                                                     'sigmoid')
    assert F.param_names(spec) == enc("") + lstm("f_blstm1") + lstm("b_blstm1") + ["softmax.W", "softmax.b"]
    assert spec["streams"][0]["enc_acts"] == ["sigmoid", "sigmoid", "sigmoid", "linear"] and spec["fusion"] == "none"


def test_adenet_v3_layer_order_matches_the_notebooks_recorded_print_network(spec_only):
    """The only record of Lasagne's ``get_all_layers`` order inside /root/reference: the print-out of ``adenet_v3`` kept in
    avletters/avletters_training.ipynb (tests/golden/adenet_v3_layers.json, made by tests/golden/make_adenet_v3_layers.py).
    The zoo module must list the same layers in the same order with the same widths -- 500-unit LSTMs for the call's
    ``lstm_size`` 250, 'output' of 26 -- and the checkpoint (pickle) order must be that list restricted to the layers that
    own parameters: raw encoder, lstm_raw, lstm_dct, diff encoder, lstm_diff, aggregation, output."""
    import json
    import os
    from ip_avsr_amd.modelzoo import adenet_v3
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adenet_v3_layers.json")))
    dims = {l["name"]: l["shape"][-1] for l in rec["layers"]}

    class Net(object):
        def __init__(self, d):
            sizes = [d, dims["fc1_raw"], dims["fc2_raw"], dims["fc3_raw"], dims["bottleneck_raw"]]
            self.layers = [None] + [_Layer(np.zeros((a, b), np.float32), np.zeros(b, np.float32)) for a, b in zip(sizes[:-1], sizes[1:])]

        def get_all_layers(self):
            return self.layers

    spec, _ = adenet_v3.create_model(Net(dims["raw_im"]), Net(dims["diff_im"]), (None, None, dims["raw_im"]), None, (None, None), None,
                                     (None, None, dims["dct"]), None, (None, None, dims["diff_im"]), None,
                                     rec["lstm_size_argument"], None, rec["output_classes_argument"], "sum")
    got = [(n, list(shape)) for n, shape in F.layer_listing(spec)]
    want = [(l["name"], l["shape"]) for l in rec["layers"]]
    assert got == want
    assert dims["lstm_raw"] == 2 * rec["lstm_size_argument"] == dims["f_lstm_agg"] and dims["output"] == 26
    # the checkpoint order = the recorded layer order, parameterised layers only
    owners = []
    for n in F.param_names(spec):
        layer = n.split(".")[0]
        if not owners or owners[-1] != layer:
            owners.append(layer)
    recorded = [l["name"] for l in rec["layers"] if l["name"].startswith(("fc", "bottleneck", "lstm", "f_lstm", "b_lstm", "output"))]
    assert owners == recorded
