"""Worker of tests/test_gpu_data_parallel.py::test_two_ranks_on_one_gpu_equal_the_single_process_run: one of N processes that
share cuda:0 (torch.distributed over gloo -- RCCL refuses two ranks on one device), each training on ITS rows of every global
minibatch, gathered on the GPU from its own resident copy of the split (DeviceSplit.batches(rank, world)).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port P tests/dp_worker_one_gpu.py OUT PRECISION
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def case():
    from oracle import adenet_oracle as O
    spec = O.spec_nstream([40, 24, 16], enc_shapes=(48, 24, 12), enc_acts=("rectify", "rectify", "linear"), lstm_size=40, classes=7,
                          fusion="concat", peepholes=True)
    rng = np.random.default_rng(12)
    p = O.init_params(spec, rng, np.float32, enc_std=0.2, perturb=0.05)
    n = 29
    lens = rng.integers(3, 12, size=n)
    total = int(lens.sum())
    streams = [rng.normal(size=(total, s["input_dim"])).astype(np.float32) for s in spec["streams"]]
    y = np.repeat(np.arange(n) % 7, lens)
    return spec, p, streams, y, lens


def train(model, split, steps, batchsize, rank=0, world=1, dp=None):
    np.random.seed(99)                                   # every rank draws the same utterance order
    gen = split.batches(batchsize, rank=rank, world=world)
    for _ in range(steps):
        b = next(gen)
        if dp is None:
            model.train_step(b.Xs, b.targets, b.mask, 2, 1e-2, want_loss=False)
        else:
            dp.train_step(b.Xs, b.targets, b.mask, 2, 1e-2, b.total_frames)
    return model.get_params_dict()


if __name__ == "__main__":
    out, precision = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from ip_avsr_amd.model import AdeNetModel
    from ip_avsr_amd.parallel import DataParallel
    from ip_avsr_amd.utils.datagen_gpu import DeviceSplit
    spec, p, streams, y, lens = case()
    spec["precision"] = precision
    model = AdeNetModel(spec)
    model.set_params_dict(p if rank == 0 else {k: np.zeros_like(v) for k, v in p.items()})
    dp = DataParallel(model)
    dp.broadcast_parameters(0)                           # ranks > 0 start from rank 0's parameters
    params = train(model, DeviceSplit(streams, y, lens), 6, 10, rank, world, dp)      # 29 utterances: 10, 10, 9 (short), 10, ...
    np.savez(os.path.join(out, "rank%d.npz" % rank), **params)
    dist.barrier()
    dist.destroy_process_group()
