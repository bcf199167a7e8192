"""Host-side functions vs golden vectors captured from the reference's own NumPy code
(tests/golden/make_golden.py imports /root/reference/utils/*.py in the build container)."""
import os

import numpy as np
import pytest

from ip_avsr_amd.utils import preprocessing as pp
from ip_avsr_amd.utils import datagen as dg
from ip_avsr_amd.utils import regularization as reg
from ip_avsr_amd.utils.data_structures import circular_list


@pytest.fixture(scope="module")
def G(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "host_golden.npz")))


def same(a, b, tol=0):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if tol:
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol)
    else:
        np.testing.assert_array_equal(a, b)


def test_deltas_known_answer_from_reference_test_delta(G):
    out = pp.deltas(G["deltas_testdelta_in"], 9)
    same(out, G["deltas_testdelta_out"])
    # SURVEY §8c known answer: rows [0,0,0,0,36,63,81,90,90] x {1,2,3,4}
    same(out, np.outer([1, 2, 3, 4], [0, 0, 0, 0, 36, 63, 81, 90, 90]))
    assert out.dtype == G["deltas_testdelta_out"].dtype


def test_deltas_random(G):
    same(pp.deltas(G["deltas_in"], 9), G["deltas_w9"], 1e-12)
    same(pp.deltas(G["deltas_in"], 5), G["deltas_w5"], 1e-12)


def test_concat_first_second_deltas(G):
    same(pp.concat_first_second_deltas(G["cfsd_in"], G["lens"], 9), G["cfsd_out_w9"], 1e-12)
    same(pp.concat_first_second_deltas(G["cfsd_in"], G["lens"], 3), G["cfsd_out_w3"], 1e-12)


def test_reorder_meanremove_diff(G):
    same(pp.reorder_data(G["X"].copy(), tuple(G["imshape"])), G["reorder_out"])
    same(pp.sequencewise_mean_image_subtraction(G["X"].copy(), G["lens"]), G["meanrm_out"], 1e-6)
    same(pp.compute_diff_images(G["X"].copy(), G["lens"]), G["diff_out"])


def test_normalisers(G):
    X = G["X"].copy()
    out = pp.normalize_input(X)
    assert out is X                                   # in place, like the reference
    same(out, G["norm_out"], 2e-6)
    fo, fm, fs = pp.featurewise_normalize_sequence(G["X"].copy())
    same(fo, G["fnorm_out"], 2e-6)
    same(fm, G["fnorm_mean"], 1e-6)
    same(fs, G["fnorm_std"], 1e-6)
    assert fo.dtype == G["fnorm_out"].dtype


def test_zigzag_and_dct(G):
    same(pp.zigzag(G["zigzag_in"]), G["zigzag_out"])
    same(pp.fill_zigzag((3, 4)), G["fill_zigzag_3x4"])
    same(pp.fill_zigzag((5, 3)), G["fill_zigzag_5x3"])
    # the reference's own test_zigzag assertion (utils/preprocessing.py:402-414)
    for X in (np.array([[1, 2, 6, 7], [3, 5, 8, 11], [4, 9, 10, 12]]),
              np.array([[1, 2, 5, 6, 9, 10], [3, 4, 7, 8, 11, 12]])):
        r = pp.zigzag(X)
        assert all(r[i] < r[i + 1] for i in range(len(r) - 1))
    same(pp.compute_dct_features(G["X"].astype("float64"), tuple(G["imshape"]), no_coeff=10),
         G["dct_out"], 1e-12)


def test_split_helpers(G):
    n = len(G["X"])
    same(pp.create_split_index(n, G["lens"], G["iters"]), G["split_index"])
    tr, te = pp.split_videolen(G["lens"], G["iters"])
    same(tr, G["split_videolen_train"])
    same(te, G["split_videolen_test"])


def test_split_seq_data_including_last_run_quirk(G):
    parts = pp.split_seq_data(G["X"], G["y"], G["subjects"], G["lens"], [1, 3], [2], [4])
    names = ["train_X", "train_y", "train_vidlens", "train_subjects", "val_X", "val_y", "val_vidlens",
             "val_subjects", "test_X", "test_y", "test_vidlens", "test_subjects"]
    for got, nm in zip(parts, names):
        same(got, G["sss_" + nm])
        assert got.dtype == G["sss_" + nm].dtype, nm


def test_split_seq_data_regular_case():
    # two utterances per subject: plain contiguous split
    lens = np.array([2, 3, 1, 2, 2, 2])
    subj = np.array([1, 1, 2, 2, 5, 5])
    X = np.arange(lens.sum() * 2, dtype="float32").reshape(-1, 2)
    y = np.repeat(np.arange(6), lens)
    p = pp.split_seq_data(X, y, subj, lens, [1], [5], [2])
    same(p[0], X[:5]); same(p[2], [2, 3]); same(p[4], X[8:]); same(p[8], X[5:8]); same(p[11], [2, 2])


def test_force_align(G):
    (a1, t1, l1), (a2, t2, l2) = pp.force_align((G["fa_s1"], G["fa_t1"], list(G["fa_l1"])),
                                                (G["fa_s2"], G["fa_t2"], list(G["fa_l2"])))
    same(a1, G["fa2_x1"]); same(t1, G["fa2_t1"]); same(l1, G["fa2_l1"])
    same(a2, G["fa2_x2"]); same(t2, G["fa2_t2"]); same(l2, G["fa2_l2"])
    assert list(l1) == list(l2)                       # what reference test/test_preprocessing.py checks


def test_multistream_force_align(G):
    streams = [(G["fa_s%d" % k], G["fa_t%d" % k], list(G["fa_l%d" % k])) for k in (1, 2, 3)]
    out = pp.multistream_force_align(streams)
    for k, (x, t, l) in enumerate(out):
        same(x, G["msfa_x%d" % k]); same(t, G["msfa_t%d" % k]); same(l, G["msfa_l%d" % k])
    assert list(out[0][2]) == list(out[1][2]) == list(out[2][2])
    assert len(out[0][0]) == len(out[1][0]) == len(out[2][0])


def test_gen_lstm_batch_random_unshuffled_with_short_last_batch(G):
    gen = dg.gen_lstm_batch_random(G["X"], G["y"], G["lens"], batchsize=4, shuffle=False)
    sizes = []
    for b in range(4):
        Xb, yb, mb, ib = next(gen)
        same(Xb, G["glbr_ns_%d_X" % b]); same(yb, G["glbr_ns_%d_y" % b])
        same(mb, G["glbr_ns_%d_mask" % b]); same(list(ib), G["glbr_ns_%d_idx" % b])
        assert yb.dtype == np.uint8 and mb.dtype == np.uint8 and Xb.dtype == G["X"].dtype
        sizes.append(len(Xb))
    assert sizes == [4, 4, 1, 4]                      # short remainder, then restart


def test_gen_lstm_batch_random_seeded_shuffle(G):
    np.random.seed(77)
    gen = dg.gen_lstm_batch_random(G["X"], G["y"], G["lens"], batchsize=4, shuffle=True)
    for b in range(4):
        Xb, yb, mb, ib = next(gen)
        same(Xb, G["glbr_sh_%d_X" % b]); same(yb, G["glbr_sh_%d_y" % b])
        same(mb, G["glbr_sh_%d_mask" % b]); same(list(ib), G["glbr_sh_%d_idx" % b])


def test_integral_len_and_gather(G):
    il = dg.compute_integral_len(G["lens"])
    same(il, G["integral_len"])
    same(dg.gen_seq_batch_from_idx(G["gsbi_data"], G["gsbi_idx"], G["lens"], il, int(G["lens"].max())),
         G["gsbi_out"])


def test_early_stop(G):
    flat, lens = G["es_windows_flat"], G["es_windows_len"]
    off = np.concatenate(([0], np.cumsum(lens)))
    es1, es2 = [], []
    for k in range(len(lens)):
        wdw = list(flat[off[k]:off[k + 1]])
        es1.append(bool(reg.early_stop(wdw)))
        for best, thr in ((1.5, 2), (2.5, 1), (0.5, 3)):
            r = reg.early_stop2(wdw, best, thr)
            es2.append(-1 if r is None else int(bool(r)))
    same(es1, G["early_stop"])
    same(es2, G["early_stop2"])


def test_circular_list(G):
    cl = circular_list(5)
    for v in range(1, 8):
        cl.push(v)
    cl[1] = 8
    same([cl[i] for i in range(len(cl))], G["circular_list_after"])
    assert [3, 8, 5, 6, 7] == list(cl) and len(cl) == 5       # reference test_circular_list asserts
    assert all(item == "hello" for item in circular_list(7, "hello"))
    assert circular_list(2).pop() is None


def test_batch_iterator_keeps_the_reference_cursor_arithmetic():
    """utils/datagen.py:311-342: remainder batch zero-padded to the batch size; ``start += end``."""
    from ip_avsr_amd.utils.datagen import batch_iterator
    np.random.seed(3)
    perm = np.random.permutation(300)
    np.random.seed(3)
    X = np.arange(1, 301, dtype=np.float32)[:, None]
    it = batch_iterator(X, X * 2, 128)
    b1, y1 = next(it); b2, _ = next(it); b3, _ = next(it)
    assert b1.shape == (128, 1) and np.array_equal(b1[:, 0], X[perm[:128], 0]) and np.array_equal(y1, 2 * b1)
    assert np.array_equal(b2[:, 0], X[perm[128:256], 0])
    assert not b3.any()                                   # start = 128 + 256 = 384 >= 300: an all-zero batch, then a new pass
    b4, _ = next(it)
    assert b4.all()


def test_resize_images_bytescale_and_antialiased_bilinear():
    from ip_avsr_amd.utils.preprocessing import resize_images, _resample_matrix
    M = _resample_matrix(80, 40)
    assert np.allclose(M.sum(1), 1) and np.allclose(M[5, 9:13], [0.125, 0.375, 0.375, 0.125])
    img = np.tile(np.linspace(0, 1, 80)[None, :], (60, 1))              # horizontal ramp, Fortran-flattened like the .mat rows
    out = resize_images(img.reshape(1, -1, order="F"), (60, 80), (30, 40)).reshape(30, 40)
    assert out.min() >= 0 and out.max() <= 255 and np.all(np.diff(out[7]) >= 0) and abs(out[7, 20] - 131) < 8
    assert np.allclose(out, out[0][None, :], atol=1.0)          # every row of a horizontal ramp (rounding to 8 bits aside)
