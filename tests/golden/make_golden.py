"""Generates tests/golden/host_golden.npz by IMPORTING the reference's host-side NumPy code
(utils/preprocessing.py, utils/datagen.py, utils/regularization.py, utils/data_structures.py)
from /root/reference and running it on small seeded inputs.

Runs only in the build container (the reference does not exist on the GPU box); the committed
.npz holds inputs and expected outputs only -- no reference source.  Import-time shims
(SURVEY.md §8c): scipy.misc.imresize was removed in SciPy>=1.3 (utils/preprocessing.py:8),
utils/io.py:3 imports lasagne, utils/preprocessing.py:431 uses xrange.

    python tests/golden/make_golden.py
"""
import builtins
import os
import sys
import types

sys.dont_write_bytecode = True          # never drop __pycache__ into the reference mount
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_golden.npz")


def import_reference():
    import scipy.misc
    if not hasattr(scipy.misc, "imresize"):
        scipy.misc.imresize = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("stub"))
    las = types.ModuleType("lasagne")
    las.layers = types.ModuleType("lasagne.layers")
    sys.modules.setdefault("lasagne", las)
    sys.modules.setdefault("lasagne.layers", las.layers)
    builtins.xrange = range
    sys.path.insert(0, REF)
    import utils.preprocessing as pp
    import utils.datagen as dg
    import utils.regularization as reg
    import utils.data_structures as ds
    return pp, dg, reg, ds


def main():
    pp, dg, reg, ds = import_reference()
    pp.test_zigzag()
    # (ds.test_circular_list() iterates with the Python-2 `next` protocol; exercise push/index instead)
    cl = ds.circular_list(5)
    for v in range(1, 8):
        cl.push(v)
    cl[1] = 8
    circ = np.array([cl[i] for i in range(len(cl))])
    rng = np.random.RandomState(1234)
    G = {'circular_list_after': circ}

    # ---- a small multi-subject dataset --------------------------------------------------
    lens = np.array([5, 3, 6, 4, 7, 3, 5, 4, 6], dtype="int64")
    subjects = np.array([1, 1, 1, 2, 2, 3, 3, 3, 4], dtype="int64")
    iters = np.array([1, 2, 3, 1, 2, 3, 1, 2, 3], dtype="int64")
    n = int(lens.sum())
    h, w = 4, 6
    X = rng.normal(size=(n, h * w)).astype("float32")
    y = np.concatenate([np.full(l, i % 5, dtype="int64") for i, l in enumerate(lens)])
    G.update(lens=lens, subjects=subjects, iters=iters, X=X, y=y, imshape=np.array([h, w]))

    # ---- utils/preprocessing.py ---------------------------------------------------------
    a = np.array([[1, 1, 1, 1, 1, 1, 1, 1, 10], [2, 2, 2, 2, 2, 2, 2, 2, 20],
                  [3, 3, 3, 3, 3, 3, 3, 3, 30], [4, 4, 4, 4, 4, 4, 4, 4, 40]])
    G["deltas_testdelta_in"] = a
    G["deltas_testdelta_out"] = pp.deltas(a, 9)                       # preprocessing.py:11-14
    xd = rng.normal(size=(3, 11))
    G["deltas_in"] = xd
    G["deltas_w9"] = pp.deltas(xd, 9)
    G["deltas_w5"] = pp.deltas(xd, 5)
    feats = rng.normal(size=(n, 3))
    G["cfsd_in"] = feats
    G["cfsd_out_w9"] = pp.concat_first_second_deltas(feats, lens, 9)
    G["cfsd_out_w3"] = pp.concat_first_second_deltas(feats, lens, 3)
    G["reorder_out"] = pp.reorder_data(X.copy(), (h, w))
    G["meanrm_out"] = pp.sequencewise_mean_image_subtraction(X.copy(), lens)
    G["diff_out"] = pp.compute_diff_images(X.copy(), lens)
    G["norm_out"] = pp.normalize_input(X.copy())
    fo, fm, fs = pp.featurewise_normalize_sequence(X.copy())
    G.update(fnorm_out=fo, fnorm_mean=fm, fnorm_std=fs)
    G["dct_out"] = pp.compute_dct_features(X.astype("float64"), (h, w), no_coeff=10, method="zigzag")
    zin = np.arange(12).reshape(3, 4)
    G["zigzag_in"] = zin
    G["zigzag_out"] = pp.zigzag(zin)
    G["fill_zigzag_3x4"] = pp.fill_zigzag((3, 4))
    G["fill_zigzag_5x3"] = pp.fill_zigzag((5, 3))
    G["split_index"] = pp.create_split_index(n, lens, iters)
    tr, te = pp.split_videolen(lens, iters)
    G["split_videolen_train"] = np.array(tr)
    G["split_videolen_test"] = np.array(te)
    parts = pp.split_seq_data(X, y, subjects, lens, [1, 3], [2], [4])
    for k, nm in enumerate(["train_X", "train_y", "train_vidlens", "train_subjects",
                            "val_X", "val_y", "val_vidlens", "val_subjects",
                            "test_X", "test_y", "test_vidlens", "test_subjects"]):
        G["sss_" + nm] = parts[k]

    # force_align / multistream_force_align (lens lists are mutated in place by the reference)
    l1 = [4, 3, 5]; l2 = [3, 5, 5]; l3 = [6, 2, 5]
    s1 = rng.normal(size=(sum(l1), 2)); s2 = rng.normal(size=(sum(l2), 3)); s3 = rng.normal(size=(sum(l3), 1))
    t1 = np.concatenate([np.full(l, i) for i, l in enumerate(l1)])
    t2 = np.concatenate([np.full(l, i) for i, l in enumerate(l2)])
    t3 = np.concatenate([np.full(l, i) for i, l in enumerate(l3)])
    G.update(fa_s1=s1, fa_s2=s2, fa_s3=s3, fa_t1=t1, fa_t2=t2, fa_t3=t3,
             fa_l1=np.array(l1), fa_l2=np.array(l2), fa_l3=np.array(l3))
    (a1, at1, al1), (a2, at2, al2) = pp.force_align((s1, t1, list(l1)), (s2, t2, list(l2)))
    G.update(fa2_x1=a1, fa2_t1=at1, fa2_l1=np.array(al1), fa2_x2=a2, fa2_t2=at2, fa2_l2=np.array(al2))
    ms = pp.multistream_force_align([(s1, t1, list(l1)), (s2, t2, list(l2)), (s3, t3, list(l3))])
    for k, (mx, mt, ml) in enumerate(ms):
        G["msfa_x%d" % k] = mx
        G["msfa_t%d" % k] = mt
        G["msfa_l%d" % k] = np.array(ml)

    # ---- utils/datagen.py ---------------------------------------------------------------
    gen = dg.gen_lstm_batch_random(X, y, lens, batchsize=4, shuffle=False)
    for b in range(4):                                   # 4,4,1(short, reset),4
        Xb, yb, mb, ib = next(gen)
        G["glbr_ns_%d_X" % b] = Xb
        G["glbr_ns_%d_y" % b] = yb
        G["glbr_ns_%d_mask" % b] = mb
        G["glbr_ns_%d_idx" % b] = np.array(list(ib))
    np.random.seed(77)
    gen = dg.gen_lstm_batch_random(X, y, lens, batchsize=4, shuffle=True)
    for b in range(4):
        Xb, yb, mb, ib = next(gen)
        G["glbr_sh_%d_X" % b] = Xb
        G["glbr_sh_%d_y" % b] = yb
        G["glbr_sh_%d_mask" % b] = mb
        G["glbr_sh_%d_idx" % b] = np.array(list(ib))
    il = dg.compute_integral_len(lens)
    G["integral_len"] = np.array(il)
    other = rng.normal(size=(n, 3)).astype("float32")
    G["gsbi_data"] = other
    G["gsbi_idx"] = np.array([7, 0, 4])
    G["gsbi_out"] = dg.gen_seq_batch_from_idx(other, [7, 0, 4], lens, il, int(lens.max()))

    # ---- utils/regularization.py --------------------------------------------------------
    windows = [[3.0, 2.0], [1.0, 2.0, 3.0], [1.0, 3.0, 2.0], [5.0], [2.0, 2.0, 2.0], [4.0, 5.0, 6.0, 7.0]]
    es1, es2 = [], []
    for wdw in windows:
        es1.append(bool(reg.early_stop(wdw)))
        for best, thr in ((1.5, 2), (2.5, 1), (0.5, 3)):
            r = reg.early_stop2(wdw, best, thr)
            es2.append(-1 if r is None else int(bool(r)))     # the reference falls off the end -> None
    G["es_windows_flat"] = np.concatenate([np.array(wd) for wd in windows])
    G["es_windows_len"] = np.array([len(wd) for wd in windows])
    G["early_stop"] = np.array(es1)
    G["early_stop2"] = np.array(es2)

    np.savez_compressed(OUT, **G)
    print("wrote", OUT, "with", len(G), "arrays,", os.path.getsize(OUT), "bytes")
    # hygiene: the reference mount must stay untouched
    for d in ("utils", "custom", "modelzoo"):
        assert not os.path.exists(os.path.join(REF, d, "__pycache__")), "bytecode leaked into reference"


if __name__ == "__main__":
    main()
