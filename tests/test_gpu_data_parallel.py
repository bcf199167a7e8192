"""The N>1 code path on ONE GPU: a single-rank ``nccl`` (= RCCL) process group drives DataParallel exactly as
bench.py / the runners do for N ranks -- zero-copy torch view of the library's flat gradient buffer, all-reduce
on it, Adam -- and must reproduce the plain single-process step bit for bit."""
import os
import socket

import numpy as np
import pytest

from oracle import adenet_oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_single_rank_nccl_dataparallel_equals_plain_step():
    import torch
    import torch.distributed as dist
    from ip_avsr_amd import _lib
    from ip_avsr_amd.model import AdeNetModel
    from ip_avsr_amd.parallel import DataParallel, wrap_flat_buffer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        spec = O.spec_nstream([12, 9, 10], enc_shapes=(14, 6), enc_acts=("rectify", "linear"), lstm_size=10, classes=5,
                              fusion="concat")
        rng = np.random.default_rng(5)
        p = O.init_params(spec, rng, np.float32, enc_std=0.3, perturb=0.1)
        B, T = 6, 9
        lens = rng.integers(3, T + 1, size=B); lens[0] = T
        mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
        xs = [torch.tensor((rng.normal(size=(B, T, s["input_dim"])) * mask[..., None]).astype(np.float32), device="cuda")
              for s in spec["streams"]]
        y = torch.tensor(np.repeat(rng.integers(0, 5, size=(B, 1)), T, axis=1).astype(np.int32), device="cuda")
        m_d = torch.tensor(mask, device="cuda")
        ref, dpm = AdeNetModel(spec), AdeNetModel(spec)
        ref.set_params_dict(p); dpm.set_params_dict(p)
        g = wrap_flat_buffer(dpm)                                  # zero-copy view of the C library's buffer
        ptr, nbytes = dpm.flat_buffer(_lib.BUF_GRAD)
        assert g.data_ptr() == ptr and g.numel() * 4 == nbytes and g.dtype == torch.float32
        dp = DataParallel(dpm)
        # [tail] + per stream its [BatchNorm | LSTM] range + one [W | b] range per encoder layer
        assert dp.overlap and len(dp.buckets) == 1 + sum(1 + len(s["enc_shapes"]) for s in spec["streams"])
        # the buckets tile the whole flat buffer (incl. the cost tail) without overlap
        cover = sorted(dp.buckets)
        assert cover[0][0] == 0 and cover[-1][1] == g.numel()
        assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        dp.broadcast_parameters(0)
        total = float(mask.sum())
        for step in range(3):
            l_ref = ref.train_step(xs, y, m_d, 2, 1e-3)
            l_dp = dp.train_step(xs, y, m_d, 2, 1e-3, total, want_loss=True)
            assert abs(l_dp - float(l_ref)) <= 1e-6 * abs(float(l_ref))
        a, b = ref.get_all_param_values(), dpm.get_all_param_values()
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)
        ref.close(); dpm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_empty_shard_and_poisoned_gradients_single_rank(precision):
    """(both modes that run the weight-stationary LSTM kernels)  (1) A rank with an EMPTY shard joins the all-reduce with zero gradients (adn_zero_grads) -- with one rank the step
    must then leave the parameters untouched apart from Adam's zero-gradient update (m = v = 0 -> no change).
    (2) A raised exchange flag in the gradient tail makes the optimiser kernel skip the update and the next host read
    report it (what every rank sees when a peer's weight-stationary LSTM exchange timed out)."""
    import torch
    import torch.distributed as dist
    from ip_avsr_amd.model import AdeNetModel
    from ip_avsr_amd.parallel import DataParallel, wrap_flat_buffer
    from ip_avsr_amd._lib import AdenetError
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        spec = dict(O.spec_nstream([12, 9], enc_shapes=(14, 6), enc_acts=("rectify", "linear"), lstm_size=10, classes=5,
                                   fusion="sum"), precision=precision)
        rng = np.random.default_rng(7)
        p = O.init_params(spec, rng, np.float32, enc_std=0.3, perturb=0.1)
        B, T = 5, 8
        mask = np.ones((B, T), np.uint8)
        xs = [(rng.normal(size=(B, T, s["input_dim"]))).astype(np.float32) for s in spec["streams"]]
        y = np.repeat(rng.integers(0, 5, size=(B, 1)), T, axis=1).astype(np.int32)
        m = AdeNetModel(spec)
        m.set_params_dict(p)
        dp = DataParallel(m)
        before = m.get_all_param_values()
        none = [x[:0] for x in xs]
        loss = dp.train_step(none, y[:0], mask[:0], 2, 1e-2, float(mask.sum()), want_loss=True)
        assert loss == 0.0
        for u, v in zip(before, m.get_all_param_values()):
            np.testing.assert_array_equal(u, v)
        # real gradients, then raise the flag by hand and step
        m.compute_grads(xs, y, mask, 2, want_loss=False)
        g = wrap_flat_buffer(m)
        g[-7] = 1.0                                            # tail[1]
        m.apply_adam(1e-2)
        with pytest.raises(AdenetError, match="skipped"):
            m.get_all_param_values()                           # first host read after the skipped update reports it
        for u, v in zip(before, m.get_all_param_values()):     # ... and nothing was updated
            np.testing.assert_array_equal(u, v)
        l0 = m.train_step(xs, y, mask, 2, 1e-2)               # the model keeps working afterwards
        assert np.isfinite(l0)
        m.close()
    finally:
        dist.destroy_process_group()


def _bucket_snapshots_equal_final(m, xs, y, m_d, theta, order, need_early):
    import torch
    from ip_avsr_amd.parallel import wrap_flat_buffer
    g = wrap_flat_buffer(m)
    buckets = m.grad_buckets()
    events = []
    for _ in buckets:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        events.append(ev)
    m.set_bucket_events([ev.cuda_event for ev in events])
    side = torch.cuda.Stream()
    end = torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        torch.cuda.synchronize()
        m.compute_grads(xs, y, m_d, theta, want_loss=False)             # enqueues the step; the events are recorded inside
        end.record()
        snaps = []
        with torch.cuda.stream(side):
            for (b, e), ev in zip(buckets, events):
                side.wait_event(ev)
                snaps.append(g[b:e].clone())
        torch.cuda.synchronize()
        early = 0
        if order != "forked":                 # the list is in completion order: an in-order communication stream never holds a
            for a, b_ in zip(events, events[1:]):                         # ready bucket behind a later one
                assert a.elapsed_time(b_) >= -0.02, ("bucket list out of completion order", order)
        for (b, e), ev, snap in zip(buckets, events, snaps):
            assert torch.equal(snap, g[b:e]), ("bucket released before it was final", order, (b, e))
            early += ev.elapsed_time(end) > 0.05                          # ms between the bucket's event and the end of the step
        assert early >= 1 or not need_early, "no bucket was released ahead of the end of the backward pass: the check would be vacuous"
    cover = sorted(buckets)                                               # the buckets tile the buffer
    assert cover[0][0] == 0 and cover[-1][1] == g.numel() and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    m.set_bucket_events([])


def _expected_bucket_count(spec):
    return 1 + sum(1 + len(s["enc_shapes"]) for s in spec["streams"])


@pytest.mark.parametrize("order", ["default", "stream_major", "ungrouped", "forked"])
def test_a_bucket_is_final_when_its_event_fires(order, monkeypatch):
    """What the overlap relies on, checked on one GPU: at the moment a bucket's event completes, that range of the gradient
    buffer already holds its FINAL values (a copy taken behind the event on a side stream equals the range after the whole
    backward pass, bit for bit) -- for the layer-major order with grouped launches and for the stream-major one
    (ADN_DP_STREAM_MAJOR), at the bench geometry so that back-propagation is still running while the copies are taken; and
    for graphs whose streams have no encoder, one encoder layer or several.  The bucket list (one bucket per stream top and
    per encoder layer) must also be in the order the events fire -- in every mode that makes back-propagation stream-major
    (ADN_DP_STREAM_MAJOR, ADN_NO_GROUPED_BACKWARD, forked streams), not only the one named after data parallel."""
    import torch
    import bench
    from ip_avsr_amd.model import AdeNetModel
    for var in ("ADN_DP_STREAM_MAJOR", "ADN_NO_GROUPED_BACKWARD", "ADN_STREAMS"):
        monkeypatch.delenv(var, raising=False)
    if order != "default":
        monkeypatch.setenv({"stream_major": "ADN_DP_STREAM_MAJOR", "ungrouped": "ADN_NO_GROUPED_BACKWARD",
                            "forked": "ADN_STREAMS"}[order], "1")
    torch.cuda.set_device(0)
    m = AdeNetModel(bench.build_spec())
    m.set_precision("bf16")
    bench.synthetic_params(m)
    xs, y, m_d, _ = bench.synthetic_batch(torch, 0, bench.B_PER_GPU, torch.device("cuda", 0))
    for _ in range(3):
        m.train_step(xs, y, m_d, bench.THETA, 2e-3, want_loss=False)
    assert len(m.grad_buckets()) == _expected_bucket_count(m.spec)
    _bucket_snapshots_equal_final(m, xs, y, m_d, bench.THETA, order, need_early=True)
    m.close()
    rng = np.random.default_rng(3)
    B, T = 300, 24
    for spec in (O.spec_nstream([40, 30, 36], enc_shapes=(64, 48, 20), enc_acts=("rectify", "rectify", "linear"), lstm_size=40,
                                classes=7, fusion="concat", has_encoder=[True, False, True]),
                 O.spec_nstream([40, 36], enc_shapes=(48,), enc_acts=("linear",), lstm_size=32, classes=7, fusion="sum")):
        for precision in ("bf16", "f32"):
            mm = AdeNetModel(dict(spec, precision=precision))
            mm.set_params_dict(O.init_params(spec, rng, np.float32, enc_std=0.2, perturb=0.1))
            assert len(mm.grad_buckets()) == _expected_bucket_count(spec)
            mask = np.ones((B, T), np.uint8)
            xs2 = [rng.normal(size=(B, T, st["input_dim"])).astype(np.float32) for st in spec["streams"]]
            y2 = np.repeat(rng.integers(0, 7, size=(B, 1)), T, axis=1).astype(np.int32)
            _bucket_snapshots_equal_final(mm, xs2, y2, mask, 2, order, need_early=False)
            mm.close()


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_exchange_error_word_is_in_bucket_0_when_its_event_fires(precision):
    """The LSTM-exchange status of this device must already sit in tail[1] of the gradient buffer when bucket 0 (which
    contains the tail) is released to the all-reduce: a word written later would be reduced as 0 on the peers, and only the
    failing rank would skip its optimiser step.  The device word is raised by hand (adn_debug_raise_exchange_error) ahead
    of compute_grads with bucket events registered; a copy of bucket 0 taken BEHIND its event on a side stream must hold
    tail[1] == 1, the optimiser must skip the update and the next host read must report it."""
    import torch
    from ip_avsr_amd import _lib
    from ip_avsr_amd._lib import AdenetError
    from ip_avsr_amd.model import AdeNetModel
    from ip_avsr_amd.parallel import wrap_flat_buffer
    torch.cuda.set_device(0)
    lib = _lib.load()
    spec = dict(O.spec_nstream([40, 36], enc_shapes=(48, 20), enc_acts=("rectify", "linear"), lstm_size=32, classes=7,
                               fusion="concat"), precision=precision)
    rng = np.random.default_rng(11)
    B, T = 64, 12
    m = AdeNetModel(spec)
    m.set_params_dict(O.init_params(spec, rng, np.float32, enc_std=0.2, perturb=0.1))
    mask = np.ones((B, T), np.uint8)
    xs = [rng.normal(size=(B, T, st["input_dim"])).astype(np.float32) for st in spec["streams"]]
    y = np.repeat(rng.integers(0, 7, size=(B, 1)), T, axis=1).astype(np.int32)
    m.train_step(xs, y, mask, 2, 1e-3)
    before = m.get_all_param_values()
    g = wrap_flat_buffer(m)
    buckets = m.grad_buckets()
    events = []
    for _ in buckets:
        ev = torch.cuda.Event()
        ev.record()
        events.append(ev)
    m.set_bucket_events([ev.cuda_event for ev in events])
    side = torch.cuda.Stream()
    try:
        for raised in (0, 1 | (3 << 4)):
            _lib.check(lib.adn_debug_raise_exchange_error(raised))
            torch.cuda.synchronize()
            m.compute_grads(xs, y, mask, 2, want_loss=False)
            b0, e0 = buckets[0]
            assert e0 == g.numel(), "bucket 0 holds the tail"
            with torch.cuda.stream(side):
                side.wait_event(events[0])
                snap = g[b0:e0].clone()
            torch.cuda.synchronize()
            assert float(snap[-7].item()) == (1.0 if raised else 0.0), "tail[1] was not final when bucket 0 was released"
            assert float(g[-7].item()) == (1.0 if raised else 0.0)
            if raised:
                _lib.check(lib.adn_debug_raise_exchange_error(0))   # (the optimiser reads tail[1], not the device word)
                m.apply_adam(1e-2)
                with pytest.raises(AdenetError, match="skipped"):
                    m.get_all_param_values()
                for u, v in zip(before, m.get_all_param_values()):
                    np.testing.assert_array_equal(u, v)
            else:
                m.apply_adam(0.0)
                before = m.get_all_param_values()
    finally:
        lib.adn_debug_raise_exchange_error(0)
        m.set_bucket_events([])
        m.close()


@pytest.mark.parametrize("precision", ["f32", "bf16", "bf16x3"])
def test_ranged_adam_equals_whole_buffer_adam_on_the_device(precision):
    """adn_adam_begin / adn_adam_range over the bucket list / adn_adam_end leaves parameters, Adam moments, the step count
    and (bf16 modes) the next step's loss exactly where adn_apply_adam leaves them -- the bf16 parameter shadow written by
    the ranged kernels included; call-sequence errors are reported."""
    import torch
    from ip_avsr_amd._lib import AdenetError
    from ip_avsr_amd.model import AdeNetModel
    torch.cuda.set_device(0)
    spec = dict(O.spec_nstream([40, 36, 30], enc_shapes=(48, 20), enc_acts=("rectify", "linear"), lstm_size=32, classes=7,
                               fusion="concat", has_encoder=[True, True, False]), precision=precision)
    rng = np.random.default_rng(21)
    p = O.init_params(spec, rng, np.float32, enc_std=0.2, perturb=0.1)
    B, T = 48, 10
    mask = np.ones((B, T), np.uint8)
    xs = [rng.normal(size=(B, T, st["input_dim"])).astype(np.float32) for st in spec["streams"]]
    y = np.repeat(rng.integers(0, 7, size=(B, 1)), T, axis=1).astype(np.int32)
    a, b = AdeNetModel(spec), AdeNetModel(spec)
    a.set_params_dict(p); b.set_params_dict(p)
    buckets = b.grad_buckets()
    with pytest.raises(AdenetError):
        b.adam_range(0, 8)                                     # no step open
    with pytest.raises(AdenetError):
        b.adam_begin(1e-3)                                     # no gradients
    from ip_avsr_amd.parallel import wrap_flat_buffer
    ga, gb = wrap_flat_buffer(a), wrap_flat_buffer(b)
    for step in range(3):
        a.compute_grads(xs, y, mask, 2, want_loss=False)
        b.zero_grads()                                         # (marks b's gradients valid) ... and b gets a's gradients:
        gb.copy_(ga)                                           # two runs of one step may differ in the last bit (atomics)
        a.apply_adam(2e-3)
        b.adam_begin(2e-3)
        if step == 1:                                          # ... or every range in one launch (what DataParallel issues)
            b.adam_ranges(buckets)
        elif step == 2:                                        # ... or some of them singly and the rest together
            b.adam_range(*buckets[0])
            b.adam_ranges(buckets[1:])
        else:
            for (lo, hi) in buckets:
                b.adam_range(lo, hi)
        b.adam_end()
    assert a.adam_step_count() == b.adam_step_count() == 3
    sa, sb = a.get_adam_state(), b.get_adam_state()
    for u, v in zip(a.get_all_param_values() + sa["m"] + sa["v"], b.get_all_param_values() + sb["m"] + sb["v"]):
        np.testing.assert_array_equal(u, v)
    la, lb = float(a.loss(xs, y, mask, 2)), float(b.loss(xs, y, mask, 2))   # (a stale bf16 shadow would show at ~1e-3)
    assert abs(la - lb) <= 2e-6 * abs(la)
    a.close(); b.close()


@pytest.mark.parametrize("precision,world", [("f32", 2), ("bf16", 3)])
def test_ranks_sharing_one_gpu_equal_the_single_process_run(tmp_path, precision, world):
    """More than one rank with the real kernels (no multi-GPU node was ever available: this is as close as one GPU gets).  N
    processes share cuda:0 under gloo; each gathers ITS rows of every global minibatch on the GPU from its own resident copy of
    the split, back-propagates with the global frame count, all-reduces bucket by bucket on a communication stream and applies
    the ranged Adam -- the weight-stationary LSTM launches of the processes run beside each other (tests/test_gpu_residency.py:
    tenants delay, they do not deadlock).  After six steps (one of them the short last batch of a pass: 9 utterances over the
    ranks) every rank holds the same parameters, equal to a single process training on the whole minibatches up to summation
    order."""
    import subprocess
    import sys
    import torch
    from tests import dp_worker_one_gpu as W
    from ip_avsr_amd.model import AdeNetModel
    from ip_avsr_amd.utils.datagen_gpu import DeviceSplit
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29540 + world
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                    "--master-port", str(port), os.path.join(root, "tests", "dp_worker_one_gpu.py"), str(tmp_path), precision],
                   check=True, cwd=root, env=env, timeout=900)
    torch.cuda.set_device(0)
    spec, p, streams, y, lens = W.case()
    spec["precision"] = precision
    m = AdeNetModel(spec)
    m.set_params_dict(p)
    single = W.train(m, DeviceSplit(streams, y, lens), 6, 10)
    m.close()
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    # (Adam turns a gradient element at noise level into a +-lr step with the noise's sign: the comparison is on the UPDATE as a
    #  whole -- relative L2 per tensor -- not on single elements; the shards' sums differ from the whole batch's in fp32
    #  summation order only)
    tol = 2e-2 if precision == "f32" else 0.2
    trained = 0
    for k, v in single.items():
        for r in range(1, world):
            np.testing.assert_array_equal(ranks[r][k], ranks[0][k], err_msg="replicas diverged: " + k)
        upd = np.linalg.norm((v - p[k]).ravel().astype(np.float64))
        if upd < 1e-6:
            continue
        trained += 1
        err = np.linalg.norm((ranks[0][k] - v).ravel().astype(np.float64))
        assert err <= tol * upd, (k, err, upd)
    assert trained >= len(single) - 6                    # the steps did train
