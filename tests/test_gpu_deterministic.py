"""Deterministic mode (adn_set_deterministic / ADN_DETERMINISTIC=1): every reduction that otherwise follows the
arrival order of float atomics -- the LSTM kernels' bias / initial-state / peephole gradient sums over utterance groups, the
column and scalar sums, the register-staged GEMMs' split-K -- runs in a fixed order.  Asserted here: the same batch gives the
same gradient BITS launch after launch and model after model, in the three arithmetics and every LSTM kernel family; a
four-step Adam trajectory ends in the same parameter bits; the mode's results stay within float-atomic noise of the default
mode's (it reorders sums, nothing else); and the epoch driver run twice from one seed prints the same costs."""
import contextlib
import io
import os

import numpy as np
import pytest

from oracle import adenet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture()
def det():
    import torch
    from ip_avsr_amd import _lib
    torch.cuda.set_device(0)
    lib = _lib.load()
    was = lib.adn_get_deterministic()
    lib.adn_set_deterministic(1)
    yield lib
    lib.adn_set_deterministic(was)


def _case(H, B, T, seed, peepholes=True, fusion="concat"):
    spec = O.spec_nstream([40, 24], enc_shapes=(48, 24, 12), enc_acts=("rectify", "rectify", "linear"), lstm_size=H, classes=7,
                          fusion=fusion, peepholes=peepholes)
    rng = np.random.default_rng(seed)
    p = O.init_params(spec, rng, np.float32, enc_std=0.2, perturb=0.05 if H <= 256 else 0.02)
    lens = rng.integers(max(1, T // 3), T + 1, size=B); lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    xs = [(rng.normal(size=(B, T, s["input_dim"])) * mask[..., None]).astype(np.float32) for s in spec["streams"]]
    y = np.repeat((np.arange(B) % 7)[:, None], T, axis=1).astype(np.int32)
    return spec, p, xs, y, mask


def _grads(spec, p, xs, y, mask, repeats=3):
    from ip_avsr_amd.model import AdeNetModel
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    out = []
    for _ in range(repeats):
        loss = m.compute_grads(xs, y, mask, 2)
        out.append((np.float32(loss), m.get_grads_dict()))
    m.close()
    return out


@pytest.mark.parametrize("precision,H,B", [("f32", 37, 70), ("bf16", 37, 70), ("bf16", 250, 200), ("bf16", 300, 70), ("bf16x3", 37, 70),
                                           ("bf16x3", 250, 200), ("bf16x3", 300, 40)])
def test_gradients_are_bit_equal_run_after_run(det, precision, H, B):
    """B = 70 / 200: three / seven 32-utterance groups (ragged last one) -- what the group sums are summed over; peepholes on;
    H = 37 / 250: four-workgroup kernels, 300: the eight-workgroup bf16 kernels and the fp32 step kernels under bf16x3"""
    spec, p, xs, y, mask = _case(H, B, 11, seed=3 * H + B)
    spec["precision"] = precision
    runs = _grads(spec, p, xs, y, mask) + _grads(spec, p, xs, y, mask, repeats=1)          # ... and a second model
    for loss, g in runs[1:]:
        assert loss == runs[0][0]
        for k in g:
            np.testing.assert_array_equal(g[k], runs[0][1][k], err_msg=k)
    # the mode reorders sums and nothing else: the default mode's gradients differ by float-atomic noise only
    det.adn_set_deterministic(0)
    loss0, g0 = _grads(spec, p, xs, y, mask, repeats=1)[0]
    det.adn_set_deterministic(1)
    assert abs(loss0 - runs[0][0]) <= 1e-6 * abs(loss0)
    for k in g0:
        scale = max(np.abs(g0[k]).max(), 1e-6)
        assert np.abs(g0[k] - runs[0][1][k]).max() <= (2e-5 if precision != "bf16" else 5e-3) * scale, k


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16"])
def test_adam_trajectories_end_in_the_same_bits(det, precision):
    from ip_avsr_amd.model import AdeNetModel
    spec, p, xs, y, mask = _case(48, 37, 9, seed=5)
    spec["precision"] = precision
    ends = []
    for _ in range(2):
        m = AdeNetModel(spec)
        m.set_params_dict(p)
        costs = [m.train_step(xs, y, mask, 2, 1e-2) for _ in range(4)]
        ends.append((costs, m.get_params_dict()))
        m.close()
    assert ends[0][0] == ends[1][0]
    for k in ends[0][1]:
        np.testing.assert_array_equal(ends[0][1][k], ends[1][1][k], err_msg=k)


def test_the_epoch_driver_repeats_itself_from_one_seed(det, tmp_path):
    """runners/nstream.py (minibatches gathered on the GPU, Adam, evaluation) twice from one seed: identical cost curves, class
    rates and trained parameters -- what an accuracy comparison between arithmetics needs to stand on"""
    from tests.test_gpu_runner import make_dataset, INI, TAIL
    from ip_avsr_amd.runners import nstream
    root = str(tmp_path)
    make_dataset(root, 3)
    ini = os.path.join(root, "run.ini")
    with open(ini, "w") as f:
        for k in (1, 2, 3):
            f.write(INI.format(k=k, root=root, reorder="False", diff="True" if k == 2 else "False"))
        f.write(TAIL.format(fusion="concat", dropout="False", root=root))
    for precision in ("f32", "bf16x3", "bf16"):
        outs = []
        for _ in range(2):
            with contextlib.redirect_stdout(io.StringIO()):
                out = nstream.main(3, ["--config", ini, "--seed", "5", "--precision", precision])
            outs.append((out["cost_train"], out["cost_val"], out["class_rate"], out["network"].get_params_dict()))
            out["network"].close()
        a, b = outs
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], precision
        for k in a[3]:
            np.testing.assert_array_equal(a[3][k], b[3][k], err_msg="%s %s" % (precision, k))
