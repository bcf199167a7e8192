"""Data-parallel training over the GPUs of one node: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL over xGMI on ROCm).  SURVEY.md §8e: utterances of a minibatch are independent
through the whole graph, so each rank runs the full step on its shard and the only exchange is ONE
all-reduce (sum) of the flat fp32 gradient buffer, followed by an identical local Adam step on every
rank (Adam state replicated, step counters in lock-step).

Exactness: the loss normaliser is the number of valid frames of the GLOBAL batch
(custom/objectives.py:32,37).  Every rank can compute it without communication, because all ranks draw
the same utterance permutation and know every utterance's length; it is passed to the library, which
normalises BEFORE back-propagation (the +-5 gate-gradient clip inside BPTT is not scale invariant).
The sum of the ranks' gradients then equals the single-GPU gradient.  The padded length T must be the
split's global maximum on every rank (the delta layer is mask-blind).

The reference has no distributed code at all; this module is new work.
"""
import os

import numpy as np

from . import _lib


class _DeviceBuffer(object):
    """Exposes a raw device allocation through ``__cuda_array_interface__`` so that torch can wrap it
    without a copy."""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (int(n_floats),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def wrap_flat_buffer(model, which=_lib.BUF_GRAD, device=None, read_only=False):
    """Zero-copy torch view (1-D float32) of one of the model's flat buffers.  ``read_only``: a view the caller only reads
    (a parameter snapshot) -- the library then does not re-derive the bf16 copies / planes of the parameters."""
    import torch
    ptr, nbytes = model.flat_buffer(which, read_only=True) if read_only else model.flat_buffer(which)
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    t = torch.as_tensor(_DeviceBuffer(ptr, nbytes // 4), device=dev)
    if t.data_ptr() != ptr:
        raise RuntimeError("torch copied the gradient buffer instead of wrapping it")
    return t


def shard_indices(batch_idxs, rank, world_size):
    """Rank r's utterances of a global minibatch: batch_idxs[r::R]."""
    return list(batch_idxs)[rank::world_size]


class DataParallel(object):
    """Wraps an ``AdeNetModel`` living on this rank's GPU.

    ``backend_tensor``: for the CPU test path (gloo, no GPU) a callable returning a host tensor stands in
    for the device gradient buffer; on a GPU box leave it None.
    """

    def __init__(self, model, process_group=None, grad_tensor=None, overlap=None):
        import torch.distributed as dist
        self.dist = dist
        self.model = model
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.grad = grad_tensor if grad_tensor is not None else wrap_flat_buffer(model)
        self._own_grad = grad_tensor           # CPU test path: the replica owns host tensors instead of device buffers
        self._own_buffers = getattr(model, "host_buffers", None)
        self._inflight = False
        self._state_serial = 0                 # bumped by everything that changes the training state (see sync_running_statistics)
        self._stats_synced_at = None
        # Overlap: the library records one HIP event per gradient bucket as soon as that bucket is final; a communication
        # stream waits on it and all-reduces the bucket while back-propagation is still running, and Adam is applied to a
        # bucket as soon as ITS reduction has landed, while later buckets are still on the wire.  Default on
        # for GPU replicas; a CPU test replica (host gradient tensor) takes the same bucket-by-bucket path without events
        # and streams when asked to (``overlap=True``).
        if overlap is None:
            overlap = grad_tensor is None and not os.environ.get("ADN_DP_NO_OVERLAP")
        self.overlap = bool(overlap) and hasattr(model, "grad_buckets")
        self.on_device = grad_tensor is None
        self.per_bucket_update = hasattr(model, "adam_range") and not os.environ.get("ADN_DP_WHOLE_BUFFER_ADAM")
        if self.overlap:
            self.buckets = model.grad_buckets()
            covered = sorted(self.buckets)
            if covered[0][0] != 0 or covered[-1][1] != self.grad.numel() or any(a[1] != b[0] for a, b in zip(covered, covered[1:])):
                raise RuntimeError("data parallel: the gradient buckets do not cover the gradient buffer exactly once")
            # buckets released at the same point of back-propagation (the streams' ranges behind one grouped launch) go out as
            # ONE grouped collective: fewer, larger launches on the communication stream (ADN_DP_NO_COALESCE=1: one each)
            groups = model.grad_bucket_groups() if hasattr(model, "grad_bucket_groups") else list(range(len(self.buckets)))
            if os.environ.get("ADN_DP_NO_COALESCE"):
                groups = list(range(len(self.buckets)))
            self.launches = []
            for k, g in enumerate(groups):
                if self.launches and groups[self.launches[-1][-1]] == g:
                    self.launches[-1].append(k)
                else:
                    self.launches.append([k])
            # Two collectives per step by default IN THE LAYER-MAJOR ORDER: everything that is final before the encoders' first
            # layer (tops, classifier, layers L-1 .. 1: 43 of 72 MB for the bench model) goes out as ONE grouped collective
            # behind layer 1's weight gradients -- layer 0's input-free backward (its weight-gradient GEMM, 0.24 ms) covers
            # it -- and layer 0 as the second.  One event, one collective and one cross-stream wait per release point cost a single rank +0.17 ms per
            # step (6 points, profiles/r03/dp_forced.txt) before any byte moves; at 8 ranks 43 MB are ~0.2 ms of xGMI time,
            # which the one layer still to run hides as well as five earlier starts did.  ADN_DP_FINE_BUCKETS=1: one
            # collective per release point (round 3's schedule).
            # The merged head waits on ONE event (its last bucket's), which stands for the whole head only where every earlier
            # bucket is final before that event in stream order: back-propagation on one stream in the layer-major order.  In
            # the stream-major orders (ADN_DP_STREAM_MAJOR, ADN_NO_GROUPED_BACKWARD, forked side streams: the bucket groups are
            # then the identity) a stream's events are recorded on its own side stream and the last launch is merely the last
            # stream's layer 0 -- there every release point keeps its own collective behind its own event.
            n_streams = len(getattr(model, "spec", {}).get("streams", [])) if isinstance(getattr(model, "spec", None), dict) else 0
            self.layer_major = list(groups) != list(range(len(self.buckets))) or n_streams == 1 or \
                bool(getattr(model, "dp_layer_major", False))
            if (len(self.launches) > 2 and self.layer_major and not os.environ.get("ADN_DP_FINE_BUCKETS")
                    and not os.environ.get("ADN_DP_NO_COALESCE")):
                head = [k for idxs in self.launches[:-1] for k in idxs]
                self.launches = [head, self.launches[-1]]
        if self.overlap and self.on_device:
            import torch
            self._torch = torch
            self.comm_stream = torch.cuda.Stream()
            # one event per LAUNCH (the last bucket of a release group; the library skips null handles): an event record costs
            # back-propagation ~6 us, and the buckets of a group become final at the same point anyway
            self.events = [None] * len(self.buckets)
            for idxs in self.launches:
                ev = torch.cuda.Event()
                ev.record()                       # materialises the underlying hipEvent_t
                self.events[idxs[-1]] = ev
            model.set_bucket_events([ev.cuda_event if ev is not None else 0 for ev in self.events])

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank ``src``'s parameters, optimiser state and Adam step count."""
        for which in (_lib.BUF_PARAM, _lib.BUF_ADAM_M, _lib.BUF_ADAM_V):
            if self._own_grad is None:
                buf = wrap_flat_buffer(self.model, which)
            elif self._own_buffers is not None:
                buf = self._own_buffers[which]
            else:
                continue                              # a test replica without host copies of its state
            self.dist.broadcast(buf, src=src, group=self.group)
        if hasattr(self.model, "adam_step_count"):
            t = self.grad.new_tensor([float(self.model.adam_step_count())])
            self.dist.broadcast(t, src=src, group=self.group)
            self.model.set_adam_step_count(int(t.item()))
        self._state_serial += 1

    def assert_quiescent(self):
        """The invariant of DESIGN.md 7 as a check: no bucket all-reduce of this replica may be outstanding when a step's
        forward pass is enqueued -- the weight-stationary LSTM launches need (nearly) every CU resident at once, and a
        collective kernel that runs beside one can leave both half-scheduled on several GPUs, each waiting for CUs the
        other holds."""
        if self._inflight:
            raise RuntimeError("data parallel: a gradient all-reduce is still outstanding; the compute stream must join the "
                               "communication stream before the next step is enqueued")

    def train_step(self, inputs, targets, mask, window, learning_rate, global_total_frames, want_loss=False, update=None):
        """One data-parallel step on this rank's shard (``inputs`` etc. may hold ZERO utterances: the rank then contributes
        zero gradients).  ``global_total_frames`` = valid frames of the whole global batch.  ``update``: what to run on the
        reduced gradients instead of ``apply_adam(learning_rate)`` -- a callable taking the model (per-layer learning
        rates: ``lambda m: m.apply_adam_vlr(lr_map)``; other update rules).  Returns the GLOBAL cost (a host float) when
        ``want_loss`` (forces a sync)."""
        self.assert_quiescent()
        self._state_serial += 1
        if len(mask) == 0:
            self.model.zero_grads()
        else:
            self.model.compute_grads(inputs, targets, mask, window, total_frames=float(global_total_frames),
                                     want_loss=False)
        ranged = self.overlap and update is None and self.per_bucket_update
        if self.overlap:
            works = []
            self._inflight = True
            if self.on_device:
                with self._torch.cuda.stream(self.comm_stream):
                    for idxs in self.launches:
                        self.comm_stream.wait_event(self.events[idxs[-1]])   # the group's buckets are final on the compute stream
                        works.append(self._reduce([self.buckets[k] for k in idxs]))
            else:
                for idxs in self.launches:
                    works.append(self._reduce([self.buckets[k] for k in idxs]))
            # Bucket 0 comes first and holds the step's status word (a rank whose LSTM exchange timed out poisons it); every
            # update kernel reads the REDUCED word, so all ranks skip -- or apply -- the step together.
            # The communication stream is in order: the second-to-last reduction done means every earlier one is.  The compute
            # stream waits for THAT one, updates all those buckets with one launch while the last group is still on the wire,
            # then waits for the last reduction and updates its buckets: two cross-stream waits and two update launches per step
            # (one wait + update per bucket measured 0.1 ms more per step on one GPU: every wait is a bubble on the compute stream).
            n_l = len(self.launches)
            phases = [list(range(n_l))] if n_l < 2 else [list(range(n_l - 1)), [n_l - 1]]
            if ranged:
                self.model.adam_begin(learning_rate)
            try:
                for phase in phases:
                    # every work of the phase is waited on: RCCL completes in order on its one stream (the waits on the earlier
                    # works are then free), but gloo runs works on several threads and may finish k ahead of k - 1
                    for li in phase:
                        works[li].wait()
                    if ranged:
                        ranges = [self.buckets[k] for li in phase for k in self.launches[li]]
                        if hasattr(self.model, "adam_ranges"):
                            self.model.adam_ranges(ranges)
                        else:
                            for b, e in ranges:
                                self.model.adam_range(b, e)
            except BaseException:
                # a failed wait / update must not leave the ranged step open (every later step would be refused) nor the
                # collectives unobserved: drain what is outstanding, roll the step counter back, then re-raise
                for w in works:
                    try:
                        w.wait()
                    except Exception:
                        pass
                if ranged:
                    self._abort_ranged_step()
                raise
            finally:
                if self.on_device:
                    self._torch.cuda.current_stream().wait_stream(self.comm_stream)
                self._inflight = False
            if ranged:
                self.model.adam_end()
        elif self.world_size > 1:
            # same stream as the model's kernels (torch's current stream): ordered after the backward pass
            self.dist.all_reduce(self.grad, op=self.dist.ReduceOp.SUM, group=self.group)
        if ranged:
            pass
        elif update is None:
            self.model.apply_adam(learning_rate)
        else:
            update(self.model)
        if want_loss:
            return float(self.grad[-8].item())
        return None

    def _abort_ranged_step(self):
        """Closes a ranged Adam step that failed half-way and takes its increment of the step counter back."""
        try:
            t = self.model.adam_step_count()
            self.model.adam_end()
            self.model.set_adam_step_count(max(0, t - 1))
        except Exception:
            pass

    def _reduce(self, ranges):
        """Sum over the ranks of the given ranges of the gradient buffer as one (grouped) asynchronous collective."""
        if len(ranges) == 1:
            b, e = ranges[0]
            return self.dist.all_reduce(self.grad[b:e], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        if not getattr(self, "_no_coalescing", False):
            try:
                from torch.distributed.distributed_c10d import _coalescing_manager
                with _coalescing_manager(group=self.group, async_ops=True) as cm:
                    for b, e in ranges:
                        self.dist.all_reduce(self.grad[b:e], op=self.dist.ReduceOp.SUM, group=self.group)
                return cm
            except (RuntimeError, NotImplementedError, AttributeError, ImportError):
                # (a backend without grouped all-reduce for these tensors -- gloo on device memory: one collective per range)
                self._no_coalescing = True
        works = [self.dist.all_reduce(self.grad[b:e], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True) for b, e in ranges]

        class _All(object):
            def wait(self_inner):
                for w in works:
                    w.wait()
        return _All()

    def sync_running_statistics(self, force=False):
        """BatchNorm running statistics (adenet_v1 / v1_1: ``streamK.bn.mean`` / ``.bn.inv_std``) are updated from each
        rank's LOCAL shard and carry no gradient, so the all-reduce never touches them and the replicas' copies drift
        apart.  This averages them over the ranks (the batch statistics themselves stay per-shard, like any
        data-parallel BatchNorm without synchronised statistics).  Called before sharded evaluation; call it before a
        checkpoint is written as well.  No-op for models without BatchNorm."""
        names = self.model.running_statistic_names() if hasattr(self.model, "running_statistic_names") else []
        if not names or self.world_size == 1:
            return 0
        # once per training state, not per evaluation call: each set_param marks the parameters dirty (the bf16 copies / planes
        # are repacked) and the average costs a host round trip; nothing changes the statistics between two optimiser steps
        # (keyed on THIS object's count of training-state changes -- train_step, broadcast_parameters, invalidate_statistics --
        #  not on the Adam counter, which an `update` callable need not advance; stamped only once the averages are written back)
        if not force and self._state_serial == self._stats_synced_at:
            return 0
        import torch
        vals = [np.asarray(self.model.get_param(n), np.float32).reshape(-1) for n in names]
        flat = torch.as_tensor(np.concatenate(vals), device=self.grad.device)
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
        flat = (flat / float(self.world_size)).cpu().numpy()
        off = 0
        for n, v in zip(names, vals):
            self.model.set_param(n, flat[off:off + v.size].reshape(np.asarray(self.model.get_param(n)).shape))
            off += v.size
        self._stats_synced_at = self._state_serial
        return len(names)

    def invalidate_statistics(self):
        """Call after changing the model's state behind this object's back (set_param, restore_params, a checkpoint load): the
        next sharded evaluation averages the BatchNorm running statistics again."""
        self._state_serial += 1

    # ------------------------------------------------------------------ sharded evaluation
    def shard(self, n):
        """Row indices of an n-row batch this rank evaluates."""
        return list(range(n))[self.rank::self.world_size]

    def gather_rows(self, local, n):
        """Reassembles an (n, ...) array from the ranks' row shards ``local`` = full[rank::world] (any trailing shape,
        identical on every rank).  One all-gather of equally padded shards."""
        import torch
        local = np.ascontiguousarray(local)
        per = -(-n // self.world_size)
        pad = np.zeros((per,) + local.shape[1:], dtype=local.dtype)
        pad[:len(local)] = local
        dev = self.grad.device
        mine = torch.as_tensor(pad, device=dev)
        parts = [torch.empty_like(mine) for _ in range(self.world_size)]
        self.dist.all_gather(parts, mine, group=self.group)
        out = np.zeros((n,) + local.shape[1:], dtype=local.dtype)
        for r, part in enumerate(parts):
            rows = len(range(r, n, self.world_size))
            out[r::self.world_size] = part.cpu().numpy()[:rows]
        return out

    def predict_sharded(self, predict, inputs, mask, window):
        """``val_fn`` on the whole batch with the utterances split over the ranks; every rank gets the full result
        (shard the held-out utterances, gather the votes).  BatchNorm running statistics are averaged over the
        ranks first, so that every row is normalised with the same statistics."""
        self.sync_running_statistics()
        n = len(mask)
        idx = self.shard(n)
        if idx:
            local = predict([x[idx] for x in inputs], mask[idx], window)
        else:
            probe = predict([x[:1] for x in inputs], mask[:1], window)       # shape / dtype of one row
            local = probe[:0]
        return self.gather_rows(local, n)

    def loss_sharded(self, loss, inputs, targets, mask, window, weights=None):
        """A cost that is a weighted mean over utterance shards (the temporal loss: weights = valid frames per shard;
        the last-timestep cross-entropy: utterances per shard), evaluated shard-wise and combined over the ranks."""
        import torch
        self.sync_running_statistics()
        n = len(mask)
        idx = self.shard(n)
        w = float(np.sum(mask[idx])) if weights is None else float(np.sum(np.asarray(weights)[idx]))
        v = float(loss([x[idx] for x in inputs], targets[idx], mask[idx], window)) if idx else 0.0
        t = torch.tensor([v * w, w], dtype=torch.float64, device=self.grad.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(t[0].item() / t[1].item())


def reduce_and_check_equal(values, group=None):
    """Debug helper: max |x_r - x_0| over ranks of a 1-D tensor (0 means replicas are in lock-step)."""
    import torch
    import torch.distributed as dist
    ref = values.clone()
    dist.broadcast(ref, src=0, group=group)
    d = (values - ref).abs().max().reshape(1)
    dist.all_reduce(d, op=dist.ReduceOp.MAX, group=group)
    return float(d.item())
