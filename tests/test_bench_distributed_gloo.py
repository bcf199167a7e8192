"""bench.py's N > 1 branch end to end on CPU: world_size 2 and 4 under gloo, the benchmark body (`bench.run`) driving an
oracle-backed replica (tests/oracle_replica.py) in place of the MI355X model.  Checks the contract of the JSON line
(whole-job value, scaling label, global batch), that the replicas stay in lock-step, and -- strong scaling -- that the
sharded job reproduces the single-process training run on the same utterances."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, THETA = 6, 7, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    from oracle import adenet_oracle as O
    spec = O.spec_nstream([7, 6], enc_shapes=(6, 4), enc_acts=("rectify", "linear"), lstm_size=5, classes=4, fusion="concat")
    p = O.init_params(spec, np.random.default_rng(3), np.float64, enc_std=0.5, perturb=0.2)
    return spec, p


def _batch(rank, n):
    rng = np.random.default_rng(100 + rank)
    lens = rng.integers(2, T + 1, size=n); lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    xs = [rng.normal(size=(n, T, d)) * mask[..., None] for d in (7, 6)]
    y = np.repeat(rng.integers(0, 4, size=(n, 1)), T, axis=1)
    return xs, y, mask, mask


def _worker(rank, world, port, scaling, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import contextlib
    import io
    import torch
    import torch.distributed as dist
    import bench
    from tests.oracle_replica import OracleReplica
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bench.B_PER_GPU, bench.T_MAX, bench.THETA, bench.LR = B, T, THETA, 1e-2
        spec, p = _problem()
        holder = {}

        def make_model():
            holder["m"] = OracleReplica(spec, p)
            return holder["m"]

        args = bench.parse_args(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--scaling", scaling, "--no-cpu-baseline"])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = bench.run(args, make_model=make_model, batch_fn=_batch, device=torch.device("cpu"), dist_backend="gloo")
        rep = holder["m"]
        flat = np.concatenate([rep.p[n].reshape(-1) for n in rep.names])
        q.put((rank, out, buf.getvalue(), flat))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,scaling", [(2, "weak"), (4, "strong")])
def test_bench_n_gt_1_branch_under_gloo(world, scaling):
    from oracle import adenet_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scaling, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    out, printed = results[0][1], results[0][2]
    assert all(r[1] is None and r[2] == "" for r in results[1:])                       # only rank 0 reports
    line = json.loads(printed.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["scaling"] == scaling and line["steps"] == 2 and line["warmup"] == 1
    assert line["config"]["global_batch"] == (B * world if scaling == "weak" else B)
    assert line["config"]["parallelism"] == "dp%d" % world and line["unit"] == "sequences/s" and line["value"] > 0
    assert abs(line["value"] - line["config"]["global_batch"] * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    # the record verifies itself: the ranks of the process group, every rank's own clock, the exchange's cost beside a local step
    d = line["distributed"]
    assert d["process_group_ranks"] == world and d["backend"] == "gloo" and d["rccl_ranks"] is None       # (nccl would say = world)
    assert len(d["per_rank_ms_per_step"]["all"]) == world
    assert d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"] and d["per_rank_ms_per_step"]["max"] <= line["ms_per_step"] * (1 + 1e-9)
    assert d["local_step_ms"] > 0 and d["exposed_allreduce_ms_per_step"] >= 0 and d["collectives_per_step"] >= 1
    assert d["allreduce_bytes_per_step"] > 0
    assert line["config"]["prewarm_steps"] == 0 and "inputs" in line["config"]
    for r in results[1:]:
        np.testing.assert_array_equal(r[3], results[0][3])                             # replicas in lock-step
    # the same job in one process: the union of the ranks' utterances, three steps (1 warm-up + 2 timed)
    spec, p = _problem()
    if scaling == "strong":
        xs, y, mask, _ = _batch(0, B)
    else:
        parts = [_batch(r, B) for r in range(world)]
        xs = [np.concatenate([pt[0][k] for pt in parts]) for k in range(2)]
        y = np.concatenate([pt[1] for pt in parts]); mask = np.concatenate([pt[2] for pt in parts])
    st = O.adam_init(p)
    for _ in range(3):
        O.train_step(spec, p, st, xs, y, mask, THETA, 1e-2)
    ref = np.concatenate([p[n].reshape(-1) for n in O.param_names(spec)])
    np.testing.assert_allclose(results[0][3], ref, rtol=0, atol=1e-11)
