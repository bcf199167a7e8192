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
