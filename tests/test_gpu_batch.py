"""Minibatch assembly on the GPU (csrc/batch.hip, utils/datagen_gpu.py) against the reference's host generators
(utils/datagen.py:92-153 gen_lstm_batch_random, :219-229 gen_seq_batch_from_idx): bit-equal batches, masks, labels and index
sequences on the reference-generated golden fixture (tests/golden/host_golden.npz) and on ragged random splits; the
short last batch and the reshuffle; data-parallel row shards; odd widths and bfloat16 residents; the runner fed from HBM
against the runner fed from the host."""
import contextlib
import io
import os

import numpy as np
import pytest

from ip_avsr_amd.utils import datagen as dg

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_golden.npz")


@pytest.fixture(scope="module")
def G():
    return np.load(GOLDEN)


def _split(streams, y, lens, **kw):
    import torch
    torch.cuda.set_device(0)
    from ip_avsr_amd.utils.datagen_gpu import DeviceSplit
    return DeviceSplit(streams, y, lens, **kw)


def _host(t):
    return t.float().cpu().numpy() if t.dtype.is_floating_point else t.cpu().numpy()


def _check(batch, Xs_ref, y_ref, mask_ref, idx_ref):
    assert list(batch.idxs) == list(idx_ref)
    for X, R in zip(batch.Xs, Xs_ref):
        got = _host(X)
        assert got.shape == R.shape and got.dtype == np.float32
        assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(R, np.float32).view(np.uint32))     # bit for bit
    T = mask_ref.shape[1]
    for host, dev, ref, dt in ((batch.y, batch.y.dev, y_ref, np.uint8), (batch.mask, batch.mask.dev, mask_ref, np.uint8),
                               (batch.targets, batch.targets.dev, np.repeat(np.asarray(y_ref).reshape(-1, 1), T, axis=-1), np.int32)):
        assert np.asarray(host).dtype == dt and np.array_equal(np.asarray(host), ref)
        assert np.array_equal(_host(dev), ref) and _host(dev).dtype == dt


def test_golden_unshuffled_batches_with_short_last_batch(G):
    """the reference's own outputs for batchsize 4 over 9 utterances, unshuffled: 4, 4, 1 (short, reset), 4"""
    sp = _split([G["X"]], G["y"], G["lens"])
    gen = sp.batches(4, shuffle=False)
    sizes = []
    for b in range(4):
        batch = next(gen)
        _check(batch, [G["glbr_ns_%d_X" % b]], G["glbr_ns_%d_y" % b], G["glbr_ns_%d_mask" % b], G["glbr_ns_%d_idx" % b])
        sizes.append(len(batch))
    assert sizes == [4, 4, 1, 4]


@pytest.mark.parametrize("prefetch", [True, False])
def test_golden_seeded_shuffle(G, prefetch):
    np.random.seed(77)
    sp = _split([G["X"]], G["y"], G["lens"])
    gen = sp.batches(4, shuffle=True, prefetch=prefetch)
    for b in range(4):
        _check(next(gen), [G["glbr_sh_%d_X" % b]], G["glbr_sh_%d_y" % b], G["glbr_sh_%d_mask" % b], G["glbr_sh_%d_idx" % b])
    # the global NumPy stream is where the reference's generator leaves it after four batches
    after = np.random.random()
    np.random.seed(77)
    ref = dg.gen_lstm_batch_random(G["X"], G["y"], G["lens"], batchsize=4, shuffle=True)
    for b in range(4):
        next(ref)
    assert after == np.random.random()


def test_golden_gather_by_index(G):
    sp = _split([G["X"], G["gsbi_data"]], G["y"], G["lens"])
    batch = sp.gather(G["gsbi_idx"])
    assert np.array_equal(_host(batch.Xs[1]), G["gsbi_out"])


def _random_split(rng, n, widths, tmin=1, tmax=17, classes=300):
    lens = rng.integers(tmin, tmax + 1, size=n)
    total = int(lens.sum())
    streams = [rng.normal(size=(total, w)).astype(np.float32) for w in widths]
    y = np.repeat(rng.integers(0, classes, size=n), lens)            # per-frame labels; > 255 wraps like the uint8 array does
    return streams, y, lens


@pytest.mark.parametrize("widths", [(1200, 90, 1200), (30,), (7, 33, 1), (1144, 1144, 26)])
@pytest.mark.parametrize("batchsize", [1, 5, 26])
def test_ragged_multistream_sequence_equals_the_host_generators(widths, batchsize):
    """three passes over a ragged split: every stream, mask, label and index of every batch equals what the host generators give
    from the same seed -- 16-, 8-, 4- and 2-byte row alignments, the short last batch, the reshuffle"""
    rng = np.random.default_rng(5)
    n = 23
    streams, y, lens = _random_split(rng, n, widths)
    sp = _split(streams, y, lens)
    np.random.seed(9)
    gen = sp.batches(batchsize)
    got = []
    steps = 3 * (n // batchsize + 1)
    for _ in range(steps):
        b = next(gen)
        got.append(([_host(x) for x in b.Xs], np.array(b.y), np.array(b.mask), list(b.idxs), _host(b.targets.dev), b.total_frames))
    np.random.seed(9)
    ref = dg.gen_lstm_batch_random(streams[0], y, lens, batchsize=batchsize)
    il = dg.compute_integral_len(lens)
    saw_short = False
    with np.errstate(over="ignore"):
        for k in range(steps):
            X1, yb, mb, ib = next(ref)
            Xs = [X1] + [dg.gen_seq_batch_from_idx(s, ib, lens, il, int(lens.max())) for s in streams[1:]]
            gx, gy, gm, gi, gt, gf = got[k]
            assert gi == list(ib)
            for a, r in zip(gx, Xs):
                assert np.array_equal(a, r)
            assert np.array_equal(gy, yb) and np.array_equal(gm, mb) and gf == float(mb.sum())
            assert np.array_equal(gt, np.repeat(yb.reshape(-1, 1), mb.shape[1], axis=-1))
            saw_short |= len(ib) != batchsize
    assert saw_short or n % batchsize == 0 or batchsize == 1


def test_prefetched_slots_are_not_overwritten_while_in_use():
    """the consumer keeps a batch while it asks for the next one only after using it: with a slow consumer on the model's stream
    (a long kernel queue) the prefetching side stream must not overwrite the slot of the batch still being read"""
    import torch
    rng = np.random.default_rng(1)
    streams, y, lens = _random_split(rng, 40, (1200, 1200), tmin=20, tmax=40, classes=26)
    sp = _split(streams, y, lens)
    np.random.seed(3)
    gen = sp.batches(8, prefetch=True)
    big = torch.randn(4096, 4096, device="cuda")
    sums = []
    for _ in range(12):
        b = next(gen)
        for _ in range(4):
            big = big @ big * 1e-4                       # keep the stream busy AHEAD of the read below
        sums.append((b.Xs[0].double().sum() + b.Xs[1].double().sum(), b.Xs[0].clone()))
    torch.cuda.synchronize()
    il = dg.compute_integral_len(lens)
    np.random.seed(3)                                    # (both generators draw lazily from the global stream)
    ref = dg.gen_lstm_batch_random(streams[0], y, lens, batchsize=8)
    for k in range(12):
        X1, yb, mb, ib = next(ref)
        X2 = dg.gen_seq_batch_from_idx(streams[1], ib, lens, il, int(lens.max()))
        assert np.array_equal(sums[k][1].cpu().numpy(), X1)
        assert abs(float(sums[k][0]) - (X1.astype(np.float64).sum() + X2.astype(np.float64).sum())) < 1e-6


def test_data_parallel_ranks_gather_only_their_rows():
    rng = np.random.default_rng(2)
    streams, y, lens = _random_split(rng, 19, (48, 30), classes=10)
    full = _split(streams, y, lens)
    world = 4
    np.random.seed(21)
    ref, g = [], full.batches(6)
    for _ in range(8):                                   # (a batch's tensors are its slot's: copied out before the slot is reused)
        b = next(g)
        ref.append(dict(global_idxs=b.global_idxs, idxs=b.idxs, total_frames=b.total_frames, Xs=[_host(x) for x in b.Xs],
                        mask=np.array(b.mask)))
    for rank in range(world):
        np.random.seed(21)
        gen = _split(streams, y, lens).batches(6, rank=rank, world=world)
        for k in range(8):
            b = next(gen)
            rows = list(range(len(ref[k]["global_idxs"])))[rank::world]
            assert list(b.global_idxs) == list(ref[k]["global_idxs"]) and list(b.idxs) == list(ref[k]["idxs"][rows])
            assert b.total_frames == ref[k]["total_frames"]            # the GLOBAL normaliser, known without communication
            for s in range(2):
                assert np.array_equal(_host(b.Xs[s]), ref[k]["Xs"][s][rows])
            assert np.array_equal(np.asarray(b.mask), ref[k]["mask"][rows])
            assert len(b) == len(rows)                                  # 3 rows of a short batch over 4 ranks: rank 3 gets none


def test_bfloat16_residents_hold_the_rounded_values_and_feed_the_model_in_place():
    import torch
    from ip_avsr_amd.model import AdeNetModel
    from oracle import adenet_oracle as O
    rng = np.random.default_rng(4)
    streams, y, lens = _random_split(rng, 12, (48, 33), tmin=3, tmax=9, classes=5)
    sp16 = _split(streams, y, lens, dtype="bfloat16")
    sp32 = _split(streams, y, lens)
    idx = [3, 0, 11, 7]
    b16, b32 = sp16.gather(idx), sp32.gather(idx)
    for a, b in zip(b16.Xs, b32.Xs):
        assert a.dtype == torch.bfloat16 and torch.equal(a, b.to(torch.bfloat16))
    # a bf16 model reads the bfloat16 batch in place and gives the bits it gives for the float32 batch (whose first GEMM rounds
    # the same way)
    spec = O.spec_nstream([48, 33], enc_shapes=(32, 16, 8), lstm_size=12, classes=5, fusion="sum")
    spec["precision"] = "bf16"
    m = AdeNetModel(spec)
    m.set_params_dict(O.init_params(spec, rng, np.float32, enc_std=0.2, perturb=0.05))
    p16 = m.predict(b16.Xs, b16.mask, 2)
    p32 = m.predict(b32.Xs, b32.mask, 2)
    assert np.array_equal(p16, p32)
    m.close()


def test_kernel_rejects_bad_arguments_and_ignores_foreign_indices():
    import ctypes as C
    import torch
    from ip_avsr_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(0)
    frames = torch.arange(6 * 4, dtype=torch.float32, device="cuda").reshape(6, 4)
    out = torch.full((2, 3, 4), -1.0, device="cuda")
    mask = torch.full((2, 3), 9, dtype=torch.uint8, device="cuda")
    offs = torch.tensor([0, 2], dtype=torch.int64, device="cuda"); lens = torch.tensor([2, 3], dtype=torch.int32, device="cuda")
    idx = torch.tensor([1, 5], dtype=torch.int32, device="cuda")            # 5: not an utterance of this split
    st = (_lib.BatchStream * 1)()
    st[0].frames, st[0].width, st[0].elem_bytes, st[0].out = frames.data_ptr(), 4, 4, out.data_ptr()
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.adn_batch_gather(st, 1, p(offs), p(lens), None, 2, p(idx), 2, 3, p(mask), None, None, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out[0], frames[2:5]) and float(out[1].abs().sum()) == 0.0 and mask.tolist() == [[1, 1, 1], [0, 0, 0]]
    st[0].elem_bytes = 3
    assert lib.adn_batch_gather(st, 1, p(offs), p(lens), None, 2, p(idx), 2, 3, p(mask), None, None, None) == _lib.ADN_ERR_INVALID
    st[0].elem_bytes = 4
    tg = torch.zeros((2, 3), dtype=torch.int32, device="cuda")
    assert lib.adn_batch_gather(st, 1, p(offs), p(lens), None, 2, p(idx), 2, 3, p(mask), p(tg), None, None) == _lib.ADN_ERR_INVALID
    assert lib.adn_batch_gather(st, 0, p(offs), p(lens), None, 2, p(idx), 2, 3, p(mask), None, None, None) == _lib.ADN_ERR_INVALID


def test_runner_fed_from_hbm_equals_runner_fed_from_the_host(tmp_path, monkeypatch):
    """the 3-stream driver twice from one seed -- minibatches assembled on the GPU (default) and by the reference's host code
    (ADN_HOST_BATCHES=1): the same batches in the same order, so the first epochs' costs agree to float-atomic noise and the
    class rates match"""
    from tests.test_gpu_runner import make_dataset, INI, TAIL
    from ip_avsr_amd.runners import nstream
    root = str(tmp_path)
    make_dataset(root, 3)
    ini = os.path.join(root, "run.ini")
    with open(ini, "w") as f:
        for k in (1, 2, 3):
            f.write(INI.format(k=k, root=root, reorder="False", diff="True" if k == 2 else "False"))
        f.write(TAIL.format(fusion="concat", dropout="False", root=root))
    outs = []
    for host in (False, True):
        if host:
            monkeypatch.setenv("ADN_HOST_BATCHES", "1")
        else:
            monkeypatch.delenv("ADN_HOST_BATCHES", raising=False)
        with contextlib.redirect_stdout(io.StringIO()):
            out = nstream.main(3, ["--config", ini, "--seed", "5"])
        outs.append(out)
    a, b = outs
    assert len(a["cost_val"]) == len(b["cost_val"]) and len(a["epoch_seconds"]) == len(a["cost_val"])
    assert np.allclose(a["cost_train"][:2], b["cost_train"][:2], rtol=2e-3) and np.allclose(a["cost_val"][:2], b["cost_val"][:2], rtol=2e-3)
    assert np.array_equal(np.asarray(a["heldout"]["y_val"]), np.asarray(b["heldout"]["y_val"]))
    assert np.array_equal(_host(a["heldout"]["X_val"][1]), b["heldout"]["X_val"][1])
    for o in outs:
        o["network"].close()


@pytest.mark.parametrize("head", ["frames", "last"])
def test_cost_and_predictions_from_one_forward_pass(head):
    """AdeNetModel.loss_and_probs (adn_loss + adn_read_probs: what the epoch loops use for the held-out split) returns exactly
    what compute_test_cost and val_fn return from two passes -- bit for bit, in every arithmetic, for both heads"""
    import torch
    from ip_avsr_amd.model import AdeNetModel
    from oracle import adenet_oracle as O
    torch.cuda.set_device(0)
    rng = np.random.default_rng(6)
    spec = O.spec_nstream([30, 22], enc_shapes=(32, 16, 8), enc_acts=("rectify", "rectify", "linear"), lstm_size=20, classes=6, fusion="sum")
    if head == "last":
        spec["head"] = "last"
    p = O.init_params(spec, rng, np.float32, enc_std=0.2, perturb=0.05)
    B, T = 13, 9
    lens = rng.integers(2, T + 1, size=B); lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    xs = [(rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32) for d in (30, 22)]
    y = np.repeat((np.arange(B) % 6)[:, None], T, axis=1).astype(np.int32)
    for precision in ("f32", "bf16x3", "bf16"):
        m = AdeNetModel(dict(spec, precision=precision))
        m.set_params_dict(p)
        cost2, probs2 = m.loss(xs, y, mask, 2), m.predict(xs, mask, 2)
        cost1, probs1 = m.loss_and_probs(xs, y, mask, 2)
        assert cost1 == cost2 and probs1.shape == probs2.shape
        np.testing.assert_array_equal(probs1, probs2)
        m.close()


def test_edge_splits_one_utterance_one_frame_and_whole_split_batches():
    """a split of ONE utterance of ONE frame (T = 1: every row is the utterance's only frame), a batch size beyond the split (every
    batch is the whole split in a fresh order, each flagged as the last of its pass -> a reshuffle per batch, as the reference's
    generator does), and the whole() form of the held-out splits: against the host generators draw for draw"""
    rng = np.random.default_rng(3)
    one = _split([rng.normal(size=(1, 5)).astype(np.float32)], np.array([4]), np.array([1]))
    b = next(one.batches(3))
    assert b.Xs[0].shape == (1, 1, 5) and b.mask.tolist() == [[1]] and b.targets.tolist() == [[4]] and list(b.idxs) == [0]
    streams, y, lens = _random_split(rng, 7, (12, 20), classes=9)
    sp = _split(streams, y, lens)
    np.random.seed(4)
    gen = sp.batches(50)
    got = [(list(next(gen).idxs)) for _ in range(3)]
    whole = sp.whole()
    np.random.seed(4)
    ref = dg.gen_lstm_batch_random(streams[0], y, lens, batchsize=50)
    want = [list(next(ref)[3]) for _ in range(3)]
    assert got == want and all(len(g) == 7 for g in got)
    X1, yb, mb, ib = next(dg.gen_lstm_batch_random(streams[0], y, lens, batchsize=len(lens)))     # (the draw whole() made)
    assert list(whole.idxs) == list(ib) and np.array_equal(_host(whole.Xs[0]), X1) and np.array_equal(np.asarray(whole.y), yb)
    il = dg.compute_integral_len(lens)
    assert np.array_equal(_host(whole.Xs[1]), dg.gen_seq_batch_from_idx(streams[1], ib, lens, il, int(lens.max())))


def test_resident_planes_gather_the_two_planes_of_the_float32_batches():
    """DeviceSplit(dtype='planes') -- the resident form of the bf16x3 / mixed arithmetic -- gathers for every stream the hi and
    the lo bfloat16 plane of exactly the batches the float32 split gathers (same indices, padding, mask, targets)."""
    import torch
    from ip_avsr_amd.model import PlaneInput
    rng = np.random.default_rng(3)
    lens = rng.integers(3, 12, size=23)
    total = int(lens.sum())
    streams = [rng.normal(size=(total, d)).astype(np.float32) for d in (24, 40)]
    y = np.repeat(np.arange(23) % 5, lens)
    from ip_avsr_amd.utils.datagen_gpu import DeviceSplit
    a, b = DeviceSplit(streams, y, lens), DeviceSplit(streams, y, lens, dtype="planes")
    def run(split):                                      # (the generators draw from np.random as they go: one after the other)
        np.random.seed(7)
        gen, out = split.batches(batchsize=6), []
        for _ in range(9):                               # two passes incl. the short last batch
            bt = next(gen)
            out.append((bt.idxs.copy(), np.asarray(bt.mask).copy(), np.asarray(bt.targets).copy(),
                        [(x.hi.clone(), x.lo.clone()) if isinstance(x, PlaneInput) else x.clone() for x in bt.Xs]))
        return out
    for (ia, ma, ta, xa), (ib, mb, tb, xb) in zip(run(a), run(b)):
        np.testing.assert_array_equal(ia, ib)
        np.testing.assert_array_equal(ma, mb)
        np.testing.assert_array_equal(ta, tb)
        for fa, pb in zip(xa, xb):
            assert isinstance(pb, tuple)
            want = PlaneInput.split(fa)
            assert torch.equal(pb[0], want.hi) and torch.equal(pb[1], want.lo)
