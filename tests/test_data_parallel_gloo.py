"""N>1 path on CPU: two processes, gloo backend, the package's DataParallel driving a replica whose
compute is the oracle (the real replica needs a GPU).  Checks the data-parallel identity of SURVEY.md §8e:
shard gradients normalised by the GLOBAL frame count and summed by one all-reduce reproduce the
single-process step exactly, and the replicas stay in lock-step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


from tests.oracle_replica import OracleReplica  # noqa: E402


def _make_problem():
    from oracle import adenet_oracle as O
    spec = O.spec_nstream([7, 6], enc_shapes=(6, 4), enc_acts=("rectify", "linear"), lstm_size=5, classes=4,
                          fusion="concat", peepholes=True)
    rng = np.random.default_rng(42)
    p = O.init_params(spec, rng, np.float64, enc_std=0.5, perturb=0.2)
    n_utt, T = 10, 7
    lens = rng.integers(2, T + 1, size=n_utt); lens[3] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    xs = [rng.normal(size=(n_utt, T, s["input_dim"])) * mask[..., None] for s in spec["streams"]]
    y = np.repeat(rng.integers(0, 4, size=(n_utt, 1)), T, axis=1)
    return spec, p, xs, y, mask


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ip_avsr_amd.parallel import DataParallel, shard_indices
        spec, p, xs, y, mask = _make_problem()
        rep = OracleReplica(spec, p)
        dp = DataParallel(rep, grad_tensor=rep.grad)
        assert dp.world_size == world and dp.rank == rank
        batch_idxs = list(range(len(mask)))                  # every rank sees the same global batch order
        mine = shard_indices(batch_idxs, rank, world)
        total = float(mask.sum())                            # known locally: all lengths are known to all ranks
        losses = []
        for step in range(3):
            losses.append(dp.train_step([x[mine] for x in xs], y[mine], mask[mine], 2, 1e-2, total, want_loss=True))
        flat = np.concatenate([rep.p[n].reshape(-1) for n in rep.names])
        q.put((rank, losses, flat))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process():
    from oracle import adenet_oracle as O
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    results.sort(key=lambda t: t[0])
    # single-process reference on the whole batch
    spec, p, xs, y, mask = _make_problem()
    st = O.adam_init(p)
    ref_losses = [O.train_step(spec, p, st, xs, y, mask, 2, 1e-2) for _ in range(3)]
    ref_flat = np.concatenate([p[n].reshape(-1) for n in O.param_names(spec)])
    for rank, losses, flat in results:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-12, atol=1e-13)      # global cost via the tail slot
        np.testing.assert_allclose(flat, ref_flat, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(results[0][2], results[1][2])                      # replicas in lock-step


def test_shard_indices_cover_the_batch_once():
    from ip_avsr_amd.parallel import shard_indices
    idx = list(np.random.default_rng(0).permutation(23))
    for world in (1, 2, 4, 8):
        shards = [shard_indices(idx, r, world) for r in range(world)]
        assert sorted(sum(shards, [])) == sorted(idx)
        assert max(map(len, shards)) - min(map(len, shards)) <= 1


def _worker_features(rank, world, port, q):
    """Empty shards, update callable, Adam step-count broadcast, sharded evaluation."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ip_avsr_amd.parallel import DataParallel
        spec, p, xs, y, mask = _make_problem()
        rep = OracleReplica(spec, p)
        if rank == 0:
            rep.set_adam_step_count(5)
        dp = DataParallel(rep, grad_tensor=rep.grad)
        dp.broadcast_parameters(0)
        t0 = rep.adam_step_count()
        # a 3-utterance remainder batch on 4 ranks: rank 3 has nothing
        sub = [0, 1, 2]
        mine = sub[rank::world]
        calls = []

        def update(model):
            calls.append(1)
            model.apply_adam(1e-2)

        loss = dp.train_step([x[mine] for x in xs], y[mine], mask[mine], 2, 1e-2, float(mask[sub].sum()), want_loss=True,
                             update=update)
        probs = dp.predict_sharded(rep.predict, xs, mask, 2)
        cost = dp.loss_sharded(rep.loss, xs, y, mask, 2)
        flat = np.concatenate([rep.p[n].reshape(-1) for n in rep.names])
        q.put((rank, t0, len(mine), len(calls), loss, probs, cost, flat))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_four_ranks_empty_shard_update_callable_and_sharded_evaluation():
    """a rank whose shard of a short minibatch is empty must still join the all-reduce (it used to raise
    'empty batch' and leave its peers hanging); per-layer update rules go through ``update=``; the Adam step count is
    broadcast with the state; evaluation shards the held-out utterances and gathers."""
    from oracle import adenet_oracle as O
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_features, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    spec, p, xs, y, mask = _make_problem()
    st = O.adam_init(p); st["t"] = 5
    sub = [0, 1, 2]
    ref_loss = O.train_step(spec, p, st, [x[sub] for x in xs], y[sub], mask[sub], 2, 1e-2)
    ref_flat = np.concatenate([p[n].reshape(-1) for n in O.param_names(spec)])
    ref_probs = O.forward(spec, p, xs, mask, 2)
    ref_cost, _, _ = O.loss_and_grads(spec, p, xs, y, mask, 2)
    assert [r[2] for r in results] == [1, 1, 1, 0]                                   # rank 3 trained on nothing
    for rank, t0, n_mine, n_calls, loss, probs, cost, flat in results:
        assert t0 == 5 and n_calls == 1
        np.testing.assert_allclose(loss, ref_loss, rtol=1e-12)
        np.testing.assert_allclose(flat, ref_flat, rtol=0, atol=1e-12)
        np.testing.assert_allclose(probs, ref_probs, rtol=0, atol=1e-12)            # every rank holds the whole result
        np.testing.assert_allclose(cost, ref_cost, rtol=1e-12)


def test_outstanding_reduction_blocks_the_next_step():
    """a step must not be enqueued while a bucket all-reduce of the replica is outstanding."""
    from ip_avsr_amd.parallel import DataParallel
    dp = DataParallel.__new__(DataParallel)
    dp._inflight = True
    with pytest.raises(RuntimeError, match="outstanding"):
        dp.assert_quiescent()
    dp._inflight = False
    dp.assert_quiescent()


def _worker_buckets(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ip_avsr_amd.parallel import DataParallel, shard_indices
        spec, p, xs, y, mask = _make_problem()
        mine = shard_indices(list(range(len(mask))), rank, world)
        total = float(mask.sum())
        out = []
        for bucketed in (True, False):
            rep = OracleReplica(spec, p)
            dp = DataParallel(rep, grad_tensor=rep.grad, overlap=bucketed)
            assert dp.overlap == bucketed and (not bucketed or len(dp.buckets) >= 3)
            losses = [dp.train_step([x[mine] for x in xs], y[mine], mask[mine], 2, 1e-2, total, want_loss=True)
                      for _ in range(3)]
            state = np.concatenate([rep.p[n].reshape(-1) for n in rep.names] +
                                   [rep.state["m"][n].reshape(-1) for n in rep.names] +
                                   [rep.state["v"][n].reshape(-1) for n in rep.names])
            out.append((losses, state, rep.adam_step_count()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_per_bucket_adam_equals_whole_buffer_adam_bit_for_bit():
    """the bucket-by-bucket path (one all-reduce per bucket, Adam on a bucket as soon as its reduction has
    landed) leaves parameters, both Adam moments and the step count exactly where one all-reduce of the whole buffer
    followed by one Adam step leaves them."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_buckets, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, (bucketed, whole) in results:
        assert bucketed[2] == whole[2] == 3
        np.testing.assert_array_equal(bucketed[0], whole[0])
        np.testing.assert_array_equal(bucketed[1], whole[1])
    np.testing.assert_array_equal(results[0][1][0][1], results[1][1][0][1])          # replicas in lock-step


def test_bucket_list_must_cover_the_gradient_buffer():
    from ip_avsr_amd.parallel import DataParallel

    class Holes(object):
        def grad_buckets(self):
            return [(8, 16), (0, 4)]

    import torch.distributed as d
    if not d.is_initialized():
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(_free_port())
        d.init_process_group("gloo", rank=0, world_size=1)
    try:
        with pytest.raises(RuntimeError, match="cover"):
            DataParallel(Holes(), grad_tensor=torch.zeros(16), overlap=True)
    finally:
        d.destroy_process_group()


class _StatsReplica(object):
    """Just the surface sync_running_statistics touches."""

    def __init__(self, rank):
        self.v = {"bn_s1.mean": np.full((1, 5), 1.0 + rank, np.float32), "bn_s1.inv_std": np.arange(5, dtype=np.float32) * (rank + 1)}

    def running_statistic_names(self):
        return list(self.v)

    def get_param(self, n):
        return self.v[n]

    def set_param(self, n, value):
        assert np.asarray(value).shape == self.v[n].shape
        self.v[n] = np.asarray(value, np.float32)


def _worker_stats(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ip_avsr_amd.parallel import DataParallel
        rep = _StatsReplica(rank)
        dp = DataParallel(rep, grad_tensor=torch.zeros(8))
        n = dp.sync_running_statistics()
        q.put((rank, n, rep.v))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_batchnorm_running_statistics_are_averaged_over_the_ranks():
    """a BatchNorm layer's running mean / inv_std are updated from each rank's shard and have no gradient;
    sync_running_statistics (called by predict_sharded / loss_sharded) makes the replicas agree on their average."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_stats, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, n, v in results:
        assert n == 2
        np.testing.assert_array_equal(v["bn_s1.mean"], np.full((1, 5), 1.5, np.float32))
        np.testing.assert_array_equal(v["bn_s1.inv_std"], np.arange(5, dtype=np.float32) * 1.5)
