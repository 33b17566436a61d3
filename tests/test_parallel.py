"""Data-parallel gradient exchange on CPU: 2 and 4 processes, ``gloo`` backend (the GPU path uses the same code with RCCL).
Checks that the bucketed SUM all-reduce + 1/world scaling reproduces the global-batch mean gradient
(MirroredStrategy semantics, apps/train_softmax_tf2_gpus.py:49) and that replicas stay bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from facenet_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [64, 640, 128, 2048, 32, 512]
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
        n_kernel = sum(sizes)
        tail = (n_kernel, n_kernel + 96)
        done_at = {i: 2 * (len(sizes) - i) for i in range(len(sizes))}
        del done_at[len(sizes) - 1]     # the softmax classifier (last layer): complete inside the loss ops, before backward starts
        buckets = parallel.make_buckets(offs, sizes, done_at, n_kernel, tail, 20, n_buckets=3)
        assert buckets[0] == (done_at[3], offs[3], n_kernel)         # first bucket: classifier + the last backbone layers, ready with the slowest
        parallel.check_buckets(buckets, n_kernel, tail)
        g = torch.Generator().manual_seed(100 + rank)
        grad = torch.randn(tail[1], generator=g)
        local = grad.clone()
        w = torch.full((tail[1],), float(rank))               # replicas start different on purpose
        parallel.broadcast_parameters([w], src=0)
        issued = []
        for (ready, lo, hi) in buckets:                        # issue order = readiness order
            issued.append(ready)
            parallel.allreduce_bucket(grad, lo, hi)
        grad *= 1.0 / world                                    # the fused optimiser's hyper[3]
        q.put((rank, local.numpy(), grad.numpy(), w.numpy(), issued))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_allreduce_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mean = sum(r[1] for r in res) / world
    for r in res:
        assert np.allclose(r[2], mean, atol=1e-6)             # every element reduced exactly once
        assert np.array_equal(r[3], np.zeros_like(r[3]))      # broadcast from rank 0
        assert r[4] == sorted(r[4])
    for r in res[1:]:
        assert np.array_equal(res[0][2], r[2])                # replicas hold bit-identical averaged gradients
