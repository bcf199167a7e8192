"""Padded-minibatch generators, mirroring the reference's ``utils/datagen.py`` for the three
functions every N-stream runner uses.  Pinned against golden
vectors captured from the reference (tests/test_host_golden.py)."""
import numpy as np


def compute_integral_len(lengths):
    """Exclusive prefix sum of utterance lengths = frame offset of each utterance
    (reference utils/datagen.py:211-216).  Returns a Python list like the reference."""
    lengths = np.asarray(lengths).reshape(-1)
    if len(lengths) == 0:
        return [0]
    return [0] + [int(v) for v in np.cumsum(lengths[:-1])]


def _pad_gather(data, idxs, seqlens, offsets, max_timesteps, dtype):
    out = np.zeros((len(idxs), max_timesteps, data.shape[-1]), dtype=dtype)
    for row, u in enumerate(idxs):
        L = int(seqlens[u])
        s = int(offsets[u])
        out[row, :L] = data[s:s + L]
    return out


def gen_seq_batch_from_idx(data, idxs, seqlens, integral_lens, max_timesteps):
    """Gather + zero-pad another stream by the utterance indices of a batch
    (reference utils/datagen.py:219-229)."""
    return _pad_gather(data, list(idxs), seqlens, integral_lens, int(max_timesteps), data.dtype)


def gen_lstm_batch_random(X, y, seqlen, batchsize=30, shuffle=True):
    """Endless generator of (X_batch (B,Tmax,D), y_batch (B,) uint8, mask (B,Tmax) uint8, idxs)
    (reference utils/datagen.py:92-153).

    Kept behaviours: Tmax is the maximum over the WHOLE split; the permutation comes from the
    global ``np.random`` stream; when ``start + batchsize >= n`` the batch is the (possibly
    short) remainder and a new permutation is drawn -- so a split whose size is a multiple of
    the batch size still ends each pass with a full batch flagged as the last; labels are
    stored as uint8."""
    seqlen = np.asarray(seqlen).reshape(-1)
    tmax = int(np.max(seqlen))
    n = len(seqlen)
    offsets = compute_integral_len(seqlen)

    def order():
        return np.random.permutation(n) if shuffle else range(n)

    perm = order()
    start = 0
    while True:
        stop = start + batchsize
        wrap = stop >= n
        idxs = perm[start:] if wrap else perm[start:stop]
        Xb = _pad_gather(X, list(idxs), seqlen, offsets, tmax, X.dtype)
        yb = np.zeros((len(idxs),), dtype="uint8")
        mask = np.zeros((len(idxs), tmax), dtype="uint8")
        for row, u in enumerate(idxs):
            yb[row] = y[offsets[u]]
            mask[row, :int(seqlen[u])] = 1
        if wrap:
            perm = order()
            start = 0
        else:
            start = stop
        yield Xb, yb, mask, idxs


def batch_iterator(X, y, batchsize=128):
    """Endless minibatch generator of the auto-encoder trainers (reference utils/datagen.py:311-342).

    Kept behaviours: one permutation per pass from the global ``np.random`` stream; the last batch of a pass is the
    remainder ZERO-PADDED to ``batchsize`` rows; and the cursor advances by ``start += end`` (not ``start = end``), so a
    pass visits rows [0,128), [128,256), [384,512), [896,1024), ... of the permutation -- the reference's arithmetic,
    reproduced on purpose (quirks are parity)."""
    start = 0
    reset = False
    randomized = np.random.permutation(len(X))
    while True:
        end = start + batchsize
        if end >= len(X):
            reset = True
            batch_idxs = randomized[start:]
        else:
            batch_idxs = randomized[start:end]
        batch_X = np.zeros((batchsize,) + X.shape[1:], dtype=X.dtype)
        batch_y = np.zeros((batchsize,) + y.shape[1:], dtype=y.dtype)
        batch_X[:len(batch_idxs)] = X[batch_idxs]
        batch_y[:len(batch_idxs)] = y[batch_idxs]
        if reset:
            randomized = np.random.permutation(len(X))
            start = 0
            reset = False
        else:
            start += end
        yield batch_X, batch_y
