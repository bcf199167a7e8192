"""Minibatch generators over splits that stay RESIDENT in HBM (SURVEY.md §8a rows H1 / H2 as GPU work).

The reference assembles every minibatch on the host -- a Python loop with one ``np.concatenate`` per utterance and stream
(``utils/datagen.py:92-153`` ``gen_lstm_batch_random``, ``:219-229`` ``gen_seq_batch_from_idx``) -- and hands float32 NumPy
arrays to ``train``.  Here each split is uploaded ONCE (float32, or bfloat16 for the bf16 arithmetic, whose first encoder
GEMM reads bfloat16 in place: ``ADN_FLAG_BF16_INPUTS``), and one ``adn_batch_gather`` launch (csrc/batch.hip) builds the
padded ``(B, Tmax, D_s)`` tensors of all streams, the mask and the repeated targets of a minibatch from an index list.

What stays on the host, because it is the reference's observable behaviour (pinned by tests/test_host_golden.py for the host
generator, and by tests/test_gpu_batch.py for this one against it):

* the utterance order: ``np.random.permutation`` from the GLOBAL NumPy stream, drawn at the same points of the consumer's
  timeline as ``gen_lstm_batch_random`` draws it (the next pass's permutation while the last batch of a pass is produced);
* the short last batch (``start + batchsize >= n`` -> remainder, then reshuffle), SURVEY App. E-9;
* ``Tmax`` = the maximum length of the WHOLE split (the delta layer is mask-blind, App. E-2);
* labels as uint8 (App. E-6).

Data parallel: a rank gathers only ``idxs[rank::world]`` (ip_avsr_amd/parallel.py ``shard_indices``); every rank draws the same
permutation, so the global valid-frame count of a batch is known everywhere without communication.

Prefetch (``prefetch=True``, off by default): batch t + 1 is gathered on a side stream while step t runs on the model's stream
(two output slots, events both ways).  Measured on MI355X (profiles/r04/epoch_bench.txt) it LOSES to the plain in-stream
gather at both ends -- B = 26: 1.292 against 1.255 ms per step in bf16, 2.90 against 2.70 in bf16x3; B = 520: 3.87 against
3.80 ms -- the gather is 5-80 us of HBM traffic, less than what the two cross-stream event waits per step cost the model's
stream, and a byte-moving kernel beside the weight-stationary LSTM launches delays the workgroups it shares CUs with.  Kept
as an option for hosts whose step is short enough to be launch-bound.
"""
import ctypes as C

import numpy as np

from .. import _lib
from .datagen import compute_integral_len


class Resident(np.ndarray):
    """A small host array (mask, targets, labels) that knows its copy in HBM: NumPy code (``np.sum(mask, axis=-1)`` in
    ``evaluate_model2``) sees the host values, ``AdeNetModel`` takes ``.dev`` and skips the upload."""

    def __new__(cls, host, dev):
        obj = np.asarray(host).view(cls)
        obj.dev = dev
        return obj

    def __array_finalize__(self, obj):
        self.dev = None                      # derived arrays (slices, reshapes) are plain host data


class Batch(object):
    """One assembled minibatch.  ``Xs``: device tensors (B, T, D_s); ``y`` (B,) uint8, ``mask`` (B, T) uint8 and ``targets``
    (B, T) int32 are ``Resident`` host arrays with their device copies; ``idxs``: the utterances of THIS rank's rows;
    ``global_idxs``: the whole batch's; ``total_frames``: valid frames of the whole (global) batch."""

    __slots__ = ("Xs", "y", "mask", "targets", "idxs", "global_idxs", "total_frames", "_ready", "_slot")

    def __len__(self):
        return len(self.idxs)


class DeviceSplit(object):
    """One split (train / val / test) of an S-stream dataset in HBM.

    ``streams``: list of (sum of lengths, D_s) arrays; ``y``: per-FRAME labels (sum of lengths,) like the reference's
    ``targetsVec``; ``seqlen``: utterance lengths.  ``dtype``: 'float32' | 'bfloat16' | 'planes' element type of the resident copies
    (bfloat16 = round-to-nearest-even of the float32 values, what the bf16 arithmetic's first GEMM would round them to
    anyway; planes = two bfloat16 matrices per stream, hi and lo: what the bf16x3 / mixed arithmetic's GEMMs read)."""

    def __init__(self, streams, y, seqlen, dtype="float32", device=None):
        import torch
        self._torch = torch
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.AdenetError("DeviceSplit needs a GPU: there is no host fallback on the product path "
                                   "(ip_avsr_amd.utils.datagen holds the reference's host generators)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.lens = np.asarray(seqlen).reshape(-1).astype(np.int64)
        self.n = len(self.lens)
        self.offsets = np.asarray(compute_integral_len(self.lens), dtype=np.int64)[:self.n]
        self.tmax = int(self.lens.max()) if self.n else 0
        total = int(self.lens.sum())
        # 'planes': every stream as its two bfloat16 planes hi = bf16(x), lo = bf16(x - hi) -- the operand form of the bf16x3 / mixed
        # arithmetic (model.PlaneInput): the bytes of float32, gathered as 2 S two-byte streams, and the model skips its split pass
        self.planes = dtype == "planes"
        self.dtype = {"float32": torch.float32, "bfloat16": torch.bfloat16, "planes": torch.bfloat16}[dtype]
        self.elem_bytes = 4 if dtype == "float32" else 2
        self.frames = []
        self.widths = []
        for k, x in enumerate(streams):
            if not (hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)):     # (a frame matrix already in HBM is used as it is)
                x = np.asarray(x)
            if x.ndim != 2 or x.shape[0] < total:
                raise ValueError("stream %d: expected (>= %d, D) frames, got %s" % (k, total, tuple(x.shape)))
            t = x[:total] if hasattr(x, "data_ptr") else torch.as_tensor(np.ascontiguousarray(x[:total], dtype=np.float32),
                                                                         device=self.device)
            if self.planes:
                t32 = t.to(torch.float32)
                hi = t32.to(torch.bfloat16)
                self.frames.append(hi.contiguous())
                self.frames.append((t32 - hi.to(torch.float32)).to(torch.bfloat16).contiguous())
                self.widths += [int(x.shape[1])] * 2
            else:
                self.frames.append(t.to(self.dtype).contiguous())
                self.widths.append(int(x.shape[1]))
        yv = np.asarray(y).reshape(-1)
        self.y_first = yv[self.offsets].astype(np.uint8) if self.n else np.zeros((0,), np.uint8)   # datagen.py:130,142
        self.d_labels = torch.as_tensor(yv[:total].astype(np.int64).astype(np.int32), device=self.device)
        self.d_offsets = torch.as_tensor(self.offsets, device=self.device)
        self.d_lens = torch.as_tensor(self.lens.astype(np.int32), device=self.device)
        self._side = None
        self._slots = {}

    # ------------------------------------------------------------------ one gather
    def _outputs(self, B, slot):
        """Output tensors of a slot, reused while B stays the same (the short last batch gets its own)."""
        torch = self._torch
        key = (slot, B)
        if key not in self._slots:
            T = self.tmax
            self._slots[key] = dict(
                Xs=[torch.empty((B, T, w), dtype=self.dtype, device=self.device) for w in self.widths],
                mask=torch.empty((B, T), dtype=torch.uint8, device=self.device),
                targets=torch.empty((B, T), dtype=torch.int32, device=self.device),
                y=torch.empty((B,), dtype=torch.uint8, device=self.device))
        return self._slots[key]

    def _launch(self, idxs, out, stream):
        torch = self._torch
        B = len(idxs)
        if B == 0:
            return
        host = torch.from_numpy(np.ascontiguousarray(idxs, dtype=np.int32)).pin_memory()
        with torch.cuda.stream(stream):
            d_idx = host.to(self.device, non_blocking=True)      # (torch's pinned-block cache keeps `host` until the copy ran)
            arr = (_lib.BatchStream * len(self.frames))()
            for k, f in enumerate(self.frames):
                arr[k].frames, arr[k].width, arr[k].elem_bytes, arr[k].out = f.data_ptr(), self.widths[k], self.elem_bytes, \
                    out["Xs"][k].data_ptr()
            _lib.check(self._lib.adn_batch_gather(arr, len(self.frames), C.c_void_p(self.d_offsets.data_ptr()),
                                                  C.c_void_p(self.d_lens.data_ptr()), C.c_void_p(self.d_labels.data_ptr()), self.n,
                                                  C.c_void_p(d_idx.data_ptr()), B, self.tmax, C.c_void_p(out["mask"].data_ptr()),
                                                  C.c_void_p(out["targets"].data_ptr()), C.c_void_p(out["y"].data_ptr()),
                                                  C.c_void_p(stream.cuda_stream)))
        out["_keep"] = (host, d_idx)             # alive until the slot is reused (the copy and the kernel are asynchronous)

    def _batch(self, global_idxs, rank, world, out):
        idxs = np.asarray(global_idxs, dtype=np.int64)
        mine = idxs[rank::world]
        b = Batch()
        T = self.tmax
        lens = self.lens[mine]
        mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
        y = self.y_first[mine]
        b.Xs = out["Xs"]
        if self.planes:                                 # (hi, lo) pairs of the 2 S gathered tensors
            from ..model import PlaneInput
            b.Xs = [PlaneInput(out["Xs"][2 * k], out["Xs"][2 * k + 1]) for k in range(len(out["Xs"]) // 2)]
        b.mask = Resident(mask, out["mask"])
        b.y = Resident(y, out["y"])
        b.targets = Resident(np.repeat(y.reshape(-1, 1), T, axis=-1).astype(np.int32), out["targets"])
        b.idxs, b.global_idxs = mine, idxs
        b.total_frames = float(self.lens[idxs].sum())
        b._ready, b._slot = None, None
        return b

    def gather(self, idxs, rank=0, world=1, slot="adhoc"):
        """The batch of utterances ``idxs`` (this rank's share of it), assembled on the current stream."""
        torch = self._torch
        idxs = np.asarray(idxs, dtype=np.int64).reshape(-1)
        if len(idxs) and (idxs.min() < 0 or idxs.max() >= self.n):
            raise IndexError("utterance index outside the split (0..%d)" % (self.n - 1))
        mine = idxs[rank::world]
        out = self._outputs(len(mine), slot)
        self._launch(mine, out, torch.cuda.current_stream())
        return self._batch(idxs, rank, world, out)

    def whole(self):
        """All utterances in order, as ``next(gen_lstm_batch_random(X, y, lens, batchsize=len(lens)))`` returns them for the
        held-out splits (runners/3stream.py:336-349): the permutation drawn for it is consumed from np.random like there."""
        order = np.random.permutation(self.n)          # gen_lstm_batch_random shuffles the held-out splits too (:117)
        b = self.gather(order, slot="whole")
        np.random.permutation(self.n)                  # ... and draws the next pass's permutation before yielding (:144-147)
        return b

    # ------------------------------------------------------------------ the endless generator
    def batches(self, batchsize=30, shuffle=True, rank=0, world=1, prefetch=False):
        """Endless generator of ``Batch`` objects in the order ``gen_lstm_batch_random(X, y, seqlen, batchsize, shuffle)``
        produces its batches (reference utils/datagen.py:92-153), the other streams gathered by the same indices
        (``gen_seq_batch_from_idx``, :219-229)."""
        torch = self._torch
        n = self.n

        def order():
            return np.random.permutation(n) if shuffle else np.arange(n)

        def plan(perm, start):
            stop = start + batchsize
            wrap = stop >= n
            return (perm[start:] if wrap else perm[start:stop]), wrap, stop

        if prefetch and self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        consumed = [None, None]                         # event: the consumer is done with the slot's tensors
        perm, start, turn, pending, last = order(), 0, 0, None, None
        while True:
            cur = torch.cuda.current_stream()
            if last is not None and prefetch:           # the consumer asked for the next batch: the previous one is free
                ev = torch.cuda.Event()
                ev.record(cur)
                consumed[last] = ev
            if pending is None:
                idxs, wrap, stop = plan(perm, start)
                slot = turn % 2
                out = self._outputs(len(idxs[rank::world]), slot)
                if prefetch:
                    if consumed[slot] is not None:
                        self._side.wait_event(consumed[slot])
                    self._launch(idxs[rank::world], out, self._side)
                    ready = torch.cuda.Event()
                    ready.record(self._side)
                else:
                    self._launch(idxs[rank::world], out, cur)
                    ready = None
            else:
                idxs, wrap, stop, slot, out, ready = pending
            # the reference draws the next pass's permutation while it produces the last batch of a pass (:143-150)
            if wrap:
                perm, start = order(), 0
            else:
                start = stop
            batch = self._batch(idxs, rank, world, out)
            if ready is not None:
                cur.wait_event(ready)
            last = slot
            turn += 1
            pending = None
            if prefetch:
                # batch t + 1 goes out now, on the side stream, into the other slot: it runs beside step t.  Its indices come
                # from the permutation in force; a reshuffle it would trigger is drawn only when that batch is handed out.
                nidxs, nwrap, nstop = plan(perm, start)
                nslot = turn % 2
                nout = self._outputs(len(nidxs[rank::world]), nslot)
                if consumed[nslot] is not None:
                    self._side.wait_event(consumed[nslot])
                self._launch(nidxs[rank::world], nout, self._side)
                nready = torch.cuda.Event()
                nready.record(self._side)
                pending = (nidxs, nwrap, nstop, nslot, nout, nready)
            yield batch
