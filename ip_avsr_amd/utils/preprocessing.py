"""Host-side feature preparation, mirroring the call surface of the reference's
``utils/preprocessing.py`` for the functions on the hot path's caller side (SURVEY.md §8a row H3
and §8f-2).  Same names, argument meaning and return shapes/dtypes; implementations are
vectorised NumPy written for this package and pinned against golden vectors captured from the
reference (tests/golden/host_golden.npz, tests/test_host_golden.py).

Reference quirks that results depend on are reproduced and called out inline.
"""
import numpy as np
import scipy.fftpack as _fft


# --------------------------------------------------------------------------- deltas
def deltas(x, w=9):
    """Linear-slope deltas of each ROW of ``x`` over a w-point window
    (reference utils/preprocessing.py:17-51).

    out[:, j] = sum_{m=-h..h} m * xx[:, h + j + m],  h = w // 2, where xx is x padded with h
    copies of a column on each side.  Quirk kept: the LEFT pad replicates
    column **1**, not column 0, and there is no sum-of-squares normaliser."""
    x = np.asarray(x)
    rows, cols = x.shape
    h = int(w) // 2
    left = np.repeat(x[:, 1:2], h, axis=1)
    right = np.repeat(x[:, -1:], h, axis=1)
    xx = np.concatenate((left, x, right), axis=1)
    dt = np.float64 if xx.dtype.kind in "iub" else np.result_type(xx.dtype, np.float32)
    out = np.zeros((rows, cols), dtype=dt)
    # scipy.signal.lfilter accumulates taps in the order win[0]*x[n] + win[1]*x[n-1] + ...;
    # follow the same order (m = h, h-1, ..., -h) so float results agree to the last bit or two
    for m in range(h, -h - 1, -1):
        if m:
            out = out + np.float32(m) * xx[:, h + m:h + m + cols]
    return out


def concat_first_second_deltas(X, vidlenvec, w=9):
    """[X | deltas | deltas-of-deltas] per utterance, rows = frames
    (reference utils/preprocessing.py:465-489).  Returns float64 like the reference."""
    X = np.asarray(X)
    F = X.shape[1]
    Y = np.zeros((X.shape[0], 3 * F))
    start = 0
    for L in vidlenvec:
        L = int(L)
        seq = X[start:start + L]
        d1 = deltas(seq.T, w)
        d2 = deltas(d1, w)
        Y[start:start + L, :F] = seq
        Y[start:start + L, F:2 * F] = d1.T
        Y[start:start + L, 2 * F:] = d2.T
        start += L
    return Y


# --------------------------------------------------------------------------- splits
def _frame_owner(vid_len_vec):
    lens = np.asarray(vid_len_vec).reshape(-1).astype(np.int64)
    return np.repeat(np.arange(len(lens)), lens), lens


def create_split_index(data_len, vid_len_vec, iter_vec):
    """Boolean per-frame index: True where the utterance's iteration is 1 or 2 (train)
    (reference utils/preprocessing.py:54-74)."""
    owner, lens = _frame_owner(vid_len_vec)
    it = np.asarray(iter_vec).reshape(-1)
    is_train = (it == 1) | (it == 2)
    idx = np.zeros((data_len,), dtype=bool)
    n = min(data_len, len(owner))
    idx[:n] = is_train[owner[:n]]
    return idx


def split_videolen(videolen_vec, iter_vec):
    """(train lengths, test lengths) lists by iteration (reference utils/preprocessing.py:77-85)."""
    tr, te = [], []
    for L, it in zip(videolen_vec, iter_vec):
        (tr if (it == 1 or it == 2) else te).append(L)
    return tr, te


def split_seq_data(X, y, subjects, video_lens, train_ids, val_ids, test_ids):
    """Subject-wise train/val/test split (reference utils/preprocessing.py:111-177).

    The reference walks the per-utterance subject vector once and flushes a run of equal
    consecutive subject ids whenever the id changes.  Two behaviours of that walk are part of
    its results and are kept:
      * a run is only recognised if the subject's utterances are contiguous;
      * if the LAST utterance starts a new run (its subject has exactly one utterance and
        differs from the one before), the *preceding* run is filed under the last utterance's
        subject id and the last utterance itself is dropped (``previous_subject = subject``
        is assigned before the flush at reference line 148-150)."""
    subjects = np.asarray(subjects).reshape(-1)
    video_lens = np.asarray(video_lens).reshape(-1)
    n = len(subjects)
    D = X.shape[1]
    buckets = {k: dict(X=[np.empty((0, D), "float32")], y=[np.empty((0,), "int")],
                       l=[np.empty((0,), "int")], s=[np.empty((0,), "int")]) for k in ("train", "val", "test")}
    frame_off = np.concatenate(([0], np.cumsum(video_lens.astype(np.int64))))

    def file_run(v0, v1, label):
        which = "train" if label in train_ids else ("val" if label in val_ids else "test")
        f0, f1 = int(frame_off[v0]), int(frame_off[v1])
        b = buckets[which]
        b["X"].append(X[f0:f1]); b["y"].append(y[f0:f1])
        b["l"].append(video_lens[v0:v1]); b["s"].append(subjects[v0:v1])

    if n:
        change = np.flatnonzero(subjects[1:] != subjects[:-1]) + 1       # run starts (besides 0)
        starts = np.concatenate(([0], change))
        ends = np.concatenate((change, [n]))
        last_is_new_run = len(starts) > 1 and starts[-1] == n - 1
        if n == 1:
            # single utterance: kept only if its subject equals the walk's initial id (1)
            if subjects[0] == 1:
                file_run(0, 1, subjects[0])
        else:
            k_end = len(starts) - (1 if last_is_new_run else 0)
            for k in range(k_end):
                label = subjects[starts[k]]
                if last_is_new_run and k == k_end - 1:
                    label = subjects[-1]
                file_run(starts[k], ends[k], label)
    out = []
    for k in ("train", "val", "test"):
        b = buckets[k]
        out += [np.concatenate(b["X"]), np.concatenate(b["y"]), np.concatenate(b["l"]), np.concatenate(b["s"])]
    return tuple(out)


# --------------------------------------------------------------------------- normalisers
def normalize_input(input, centralize=True, quantize=False):
    """Per-frame (row) z-normalisation, population std, IN PLACE
    (reference utils/preprocessing.py:218-242)."""
    if centralize:
        mu = input.mean(axis=1, keepdims=True)
        c = input - mu
        input[...] = c / np.std(c, axis=1, keepdims=True)
    if quantize:
        lo = input.min(axis=1, keepdims=True)
        hi = input.max(axis=1, keepdims=True)
        input[...] = (input - lo) / (hi - lo)
    return input


def featurewise_normalize_sequence(input):
    """Column z-normalisation; returns (normalised, mean, std) so that the train statistics can
    be applied to val/test (reference utils/preprocessing.py:245-257, runners/3stream.py:102-108)."""
    mean = np.mean(input, axis=0)
    centred = input - mean
    std = np.std(centred, axis=0)
    return centred / std, mean, std


def sequencewise_mean_image_subtraction(input, seqlens, axis=0):
    """Subtract each utterance's mean frame (reference utils/preprocessing.py:260-277)."""
    out = np.zeros(input.shape, input.dtype)
    start = 0
    for L in seqlens:
        L = int(L)
        seq = input[start:start + L]
        out[start:start + L] = seq - np.sum(seq, axis, input.dtype) / L
        start += L
    return out


def compute_diff_images(X, vidlenvec):
    """Frame differences per utterance; frame 0 receives a copy of the first difference
    (reference utils/preprocessing.py:506-517)."""
    out = np.zeros(X.shape, dtype=X.dtype)
    start = 0
    for L in vidlenvec:
        L = int(L)
        d = X[start + 1:start + L] - X[start:start + L - 1]
        out[start + 1:start + L] = d
        out[start] = d[0]
        start += L
    return out


def reorder_data(X, shape, orig_order="f", desired_order="c"):
    """Re-pack flattened (d1,d2) images between Fortran and C pixel order
    (reference utils/preprocessing.py:492-503)."""
    d1, d2 = shape
    if orig_order.lower() == desired_order.lower():
        return X.reshape((-1, d1 * d2))
    n = X.reshape((-1, d1 * d2)).shape[0]
    # Reference semantics: ``X.reshape((-1,d1,d2), order=orig)`` applies the order to ALL THREE axes
    # (the image index too), then flattens with the desired order.
    cube = X.reshape((-1, d1, d2), order=orig_order.upper())
    return cube.reshape((n, d1 * d2), order=desired_order.upper())


# --------------------------------------------------------------------------- DCT / zig-zag
def _zigzag_order(rows, cols):
    """(row, col) visiting order of the reference's zig-zag walk
    (utils/preprocessing.py:280-337): anti-diagonals, odd ones walked downwards."""
    order = []
    for s in range(rows + cols - 1):
        r_lo, r_hi = max(0, s - cols + 1), min(rows - 1, s)
        rr = range(r_lo, r_hi + 1) if s % 2 else range(r_hi, r_lo - 1, -1)
        order.extend((r, s - r) for r in rr)
    return order


def zigzag(X):
    X = np.asarray(X)
    r, c = zip(*_zigzag_order(*X.shape))
    return X[np.array(r), np.array(c)]


def fill_zigzag(shape):
    out = np.zeros(shape, dtype=int)
    for i, (r, c) in enumerate(_zigzag_order(*shape)):
        out[r, c] = i + 1
    return out


def compute_dct_features(X, image_shape, no_coeff=30, method="zigzag"):
    """DCT features (reference utils/preprocessing.py:417-462).  Quirk kept: a
    **1-D** orthonormal DCT-II over the flattened image, then reshape + zig-zag, skipping DC."""
    X_dct = _fft.dct(X, norm="ortho")
    if method == "zigzag":
        r, c = zip(*_zigzag_order(*image_shape)[1:no_coeff + 1])
        flat_idx = np.array(r) * image_shape[1] + np.array(c)
        return X_dct[:, flat_idx]
    X_dct = X_dct[:, 1:]
    if method == "rel_variance":
        score = np.std(X_dct - np.mean(X_dct, 0), 0)
    elif method == "variance":
        score = np.std(X_dct, 0)
    elif method == "energy":
        score = np.sum(np.abs(X_dct), 0)
    else:
        raise NotImplementedError("method not implemented, use only 'zigzag', 'variance', 'rel_variance")
    return X_dct[:, np.argsort(score)[::-1][:no_coeff]]


# --------------------------------------------------------------------------- force alignment
def force_align(x1, x2, mode="fill"):
    """Pad the shorter of two streams, utterance by utterance, with copies of a trailing frame
    (reference utils/preprocessing.py:607-660).  Length lists are updated in place.
    Quirk kept: when stream 2 is the shorter one the filler frame is
    ``x2[start2 + l1 - 1]`` (indexed with stream 1's length, reference line 643)."""
    X1, T1, L1 = x1
    X2, T2, L2 = x2
    o1, ot1, o2, ot2 = [], [], [], []
    p1 = p2 = 0
    for i in range(len(L1)):
        l1, l2 = int(L1[i]), int(L2[i])
        if mode == "fill":
            o1.extend(X1[p1:p1 + l1]); ot1.extend(T1[p1:p1 + l1])
            o2.extend(X2[p2:p2 + l2]); ot2.extend(T2[p2:p2 + l2])
            gap = l1 - l2
            if gap < 0:
                o1.extend(np.copy(X1[p1 + l1 - 1]) for _ in range(-gap))
                ot1.extend(np.copy(T1[p1 + l1 - 1]) for _ in range(-gap))
                L1[i] = l2
            else:
                if gap:
                    filler = X2[p2 + l1 - 1]
                    o2.extend(np.copy(filler) for _ in range(gap))
                    ot2.extend(np.copy(T2[p2 + l2 - 1]) for _ in range(gap))
                L2[i] = l1
            p1 += l1
            p2 += l2
    return (np.array(o1), np.array(ot1), L1), (np.array(o2), np.array(ot2), L2)


def multistream_force_align(orig_streams, mode="fill"):
    """Align S streams to the per-utterance maximum length by repeating each stream's last frame
    (reference utils/preprocessing.py:673-712).  ``orig_streams`` = [(X, targets, lens), ...];
    the lens containers are updated in place, as in the reference."""
    S = len(orig_streams)
    lens = [s[2] for s in orig_streams]
    n = len(lens[0])
    out_x = [[] for _ in range(S)]
    out_t = [[] for _ in range(S)]
    pos = [0] * S
    for i in range(n):
        cur = [int(l[i]) for l in lens]
        longest = cur[int(np.argmax(cur))]
        for j, (X, T, _) in enumerate(orig_streams):
            l = cur[j]
            out_x[j].extend(X[pos[j]:pos[j] + l])
            out_t[j].extend(T[pos[j]:pos[j] + l])
            for _ in range(longest - l):
                out_x[j].append(np.copy(X[pos[j] + l - 1]))
                out_t[j].append(np.copy(T[pos[j] + l - 1]))
            lens[j][i] = longest
            pos[j] += l
    return [(np.array(out_x[j]), np.array(out_t[j]), lens[j]) for j in range(S)]


def _resample_matrix(n_in, n_out):
    """Row-stochastic (n_out, n_in) matrix of PIL's BILINEAR resampling: a triangle filter whose support grows with the
    down-scaling factor (anti-aliasing), centres at (i + 0.5) * n_in / n_out."""
    scale = float(n_in) / n_out
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    M = np.zeros((n_out, n_in))
    for i in range(n_out):
        center = (i + 0.5) * scale
        lo, hi = max(0, int(center - support + 0.5)), min(n_in, int(center + support + 0.5))
        w = np.maximum(0.0, 1.0 - np.abs((np.arange(lo, hi) + 0.5 - center) / fscale))
        M[i, lo:hi] = w / w.sum()
    return M


def resize_img(img, orig_dim=(60, 80), dim=(30, 40), reshape=True, order='F'):
    """``scipy.misc.imresize(img, dim)`` of the reference (utils/preprocessing.py:180-192): the image is byte-scaled
    (min -> 0, max -> 255, uint8) and resampled bilinearly the way PIL does.  scipy.misc.imresize and PIL are
    third-party code absent from this image; this restates their documented behaviour (values may differ from the
    original in the last unit of the 8-bit result -- there is no fixture of it in the reference)."""
    img = np.asarray(img, dtype=np.float64)
    if reshape:
        img = img.reshape(orig_dim, order=order)
    lo, hi = img.min(), img.max()
    span = hi - lo if hi > lo else 1.0
    b = np.clip((img - lo) * (255.0 / span) + 0.5, 0, 255).astype(np.uint8).astype(np.float64)       # bytescale
    out = _resample_matrix(b.shape[0], dim[0]) @ b @ _resample_matrix(b.shape[1], dim[1]).T
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def resize_images(images, orig_dim=(60, 80), dim=(30, 40), reshape=True, order='F'):
    """Resizes every row image of a data matrix; flattened results are in C order (reference :195-216)."""
    resized = np.zeros((images.shape[0], dim[0] * dim[1])) if reshape else np.zeros((images.shape[0], dim[0], dim[1]))
    for i, img in enumerate(images):
        r = resize_img(img, orig_dim, dim, reshape, order)
        resized[i] = r.reshape((dim[0] * dim[1],)) if reshape else r
    return resized
