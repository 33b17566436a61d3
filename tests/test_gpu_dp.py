"""Data-parallel training step on the GPU box: 2 ranks (gloo; both on cuda:0 because RCCL refuses two ranks per device),
per-replica BatchNorm, bucketed gradient all-reduce between captured/eager schedule segments, 1/world inside the optimiser.
Checks MirroredStrategy semantics (apps/train_softmax_tf2_gpus.py:49): replicas end bit-identical, and equal (up to the
training noise floor) to a single process that averages the two replicas' gradients itself."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, q, use_graph):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from facenet_amd.engine import Network
        from facenet_amd.train import Trainer
        from oracle import facenet_oracle as fo
        from tests.util_data import structured_images
        params, _, _ = fo.build_params(128, seed=0)
        net = Network(embedding_size=128, device="cuda:0", train_dtype=torch.float16)
        net.load_keras_params(params)
        tr = Trainer(net, batch=6, loss="triplet", alpha=0.2, lr=0.01, world_size=world, process_group=dist.group.WORLD, n_buckets=4)
        assert len(tr.buckets) >= 3 and len(tr.segments) == len(tr.buckets) + 1
        tr.set_images(torch.from_numpy(structured_images(6, seed=20 + rank)))
        if use_graph:
            tr.capture()                       # one eager warm-up step + capture: restart from the same state
            net.load_keras_params(params)
            tr.reset_optimizer(lr=0.01)
        tr.step()
        torch.cuda.synchronize()
        q.put((rank, net.P.cpu().numpy(), tr.G.cpu().numpy(), tr.loss_value()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_replicas_match_manual_gradient_average(use_graph):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q, use_graph)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # replicas: identical summed gradients, identical parameters (bitwise)
    assert np.array_equal(res[0][2], res[1][2])
    assert np.array_equal(res[0][1], res[1][1])
    # single-process emulation: the two replicas' forward / backward passes one after the other (the launches a world-1 trainer
    # issues, optimiser left out), gradients summed by hand, one optimiser step on their mean.  Every accumulation is order
    # independent (fixed-point statistics and bias gradients, slab-reduced weight gradients), so the data-parallel step must
    # reproduce this BIT FOR BIT: summed gradients and parameters.
    from facenet_amd.engine import Network
    from facenet_amd.train import Trainer
    from oracle import facenet_oracle as fo
    from tests.util_data import structured_images
    params, _, _ = fo.build_params(128, seed=0)
    grads = []
    for r in range(world):
        net = Network(embedding_size=128, device="cuda:0", train_dtype=torch.float16)
        net.load_keras_params(params)
        tr = Trainer(net, batch=6, loss="triplet", alpha=0.2, lr=0.01)
        tr.set_images(torch.from_numpy(structured_images(6, seed=20 + r)))
        tr.plan.run_ops([op for op in tr.step_ops if op.name not in ("adam_tick", "adam_keras", "pack_transpose")], net.stream())
        torch.cuda.synchronize()
        grads.append(tr.G.clone())
    p0 = net.P.clone()
    gsum = grads[0] + grads[1]
    tr.G.copy_(gsum / world)                     # the fused optimiser applies 1/world to the all-reduced SUM: the same product
    tr.plan.run_ops(tr.opt_ops, net.stream())
    torch.cuda.synchronize()
    assert float(gsum.abs().max()) > 0 and float((net.P - p0).abs().max()) > 0
    assert np.array_equal(res[0][2], gsum.cpu().numpy())          # all-reduced gradients == hand-summed gradients, bitwise
    assert np.array_equal(res[0][1], net.P.cpu().numpy())         # and so are the parameters after the step


def _rank_softmax(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from facenet_amd.engine import Network
        from facenet_amd.train import Trainer
        from tests.util_data import structured_images
        # every rank builds its network from a DIFFERENT seed: Trainer must broadcast rank 0's variables (MirroredStrategy)
        net = Network(embedding_size=128, device="cuda:0", train_dtype=torch.float16, nrof_classes=19, seed=rank)
        p_init = net.P.clone()
        tr = Trainer(net, batch=4, loss="softmax", lr=0.01, world_size=world, process_group=dist.group.WORLD, n_buckets=4)
        start = net.P.cpu().numpy().copy()
        # the classifier's weight gradient is complete before backward starts: its bucket is ready at launch 0 of the backward list
        assert tr.buckets[0][2] == net.n_kernel and len(tr.segments) == len(tr.buckets) + 1
        tr.set_images(torch.from_numpy(structured_images(4, seed=40 + rank)), torch.tensor([(3 * rank + i) % 19 for i in range(4)]))
        tr.capture()
        for _ in range(2):
            tr.step()
        torch.cuda.synchronize()
        mean, var = tr.averaged_moving_stats()
        q.put((rank, start, net.P.cpu().numpy(), net.S_mean.cpu().numpy(), mean.cpu().numpy(), bool(torch.equal(p_init, net.P) or rank == 0)))
    finally:
        dist.destroy_process_group()


def test_four_replicas_softmax_broadcast_buckets_and_stat_averaging():
    """World 4 (gloo, one GPU): rank 0's variables are broadcast at construction, the classifier bucket that is ready before
    backward is exchanged, replicas stay bit-identical through captured steps, per-replica moving statistics differ and
    averaged_moving_stats() returns their mean on every rank."""
    import torch.multiprocessing as mp
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_softmax, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1])          # broadcast: every replica starts from rank 0's parameters
        assert np.array_equal(res[0][2], r[2])          # and stays bit-identical after two captured steps
        assert np.array_equal(res[0][4], r[4])          # the averaged moving statistics agree on every rank
    assert not np.array_equal(res[0][3], res[1][3])     # per-replica BatchNorm: local moving statistics differ
    assert np.allclose(res[0][4], sum(r[3] for r in res) / world, rtol=1e-6, atol=1e-7)
    assert float(np.abs(res[0][2] - res[0][1]).max()) > 0


def _one_rank_rccl(port, q, use_graph):
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)   # "nccl" IS RCCL on ROCm
    try:
        from facenet_amd.engine import Network
        from facenet_amd.train import Trainer
        from oracle import facenet_oracle as fo
        from tests.util_data import structured_images
        params, _, _ = fo.build_params(128, seed=0)
        net = Network(embedding_size=128, device="cuda:0", train_dtype=torch.float16)
        net.load_keras_params(params)
        tr = Trainer(net, batch=6, loss="triplet", alpha=0.2, lr=0.01, world_size=1, process_group=dist.group.WORLD, n_buckets=4)
        assert tr.exchange and tr.comm_stream is not None and len(tr.segments) == len(tr.buckets) + 1
        tr.set_images(torch.from_numpy(structured_images(6, seed=20)))
        if use_graph:
            tr.capture()
            net.load_keras_params(params)
            tr.reset_optimizer(lr=0.01)
        for _ in range(2):
            tr.step()
        torch.cuda.synchronize()
        P, G, loss = net.P.cpu().numpy(), tr.G.cpu().numpy(), tr.loss_value()
        prof = tr.exchange_profile(steps=1)        # (a third step: after the state was read)
        q.put((P, G, loss, len(prof["buckets"]), dist.get_backend()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_one_rank_rccl_exchange_is_the_identity(use_graph):
    """The data-parallel step with its bucket all-reduces going through RCCL -- a ONE-rank communicator, all a one-GPU box allows
    (RCCL refuses two ranks on a device): communicator set-up, every `all_reduce` on the communication stream, its ordering against
    the eager / captured schedule segments and the profile's events are the real ones; the data must come back unchanged, so two
    steps end BITWISE where an ordinary single-process trainer ends."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_rccl, args=(_free_port(), q, use_graph))
    p.start()
    P, G, loss, n_buckets, backend = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert backend == "nccl" and n_buckets >= 3
    from facenet_amd.engine import Network
    from facenet_amd.train import Trainer
    from oracle import facenet_oracle as fo
    from tests.util_data import structured_images
    params, _, _ = fo.build_params(128, seed=0)
    net = Network(embedding_size=128, device="cuda:0", train_dtype=torch.float16)
    net.load_keras_params(params)
    tr = Trainer(net, batch=6, loss="triplet", alpha=0.2, lr=0.01)
    tr.set_images(torch.from_numpy(structured_images(6, seed=20)))
    p_start = net.P.clone()
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    assert float((net.P - p_start).abs().max()) > 0
    assert np.array_equal(G, tr.G.cpu().numpy())
    assert np.array_equal(P, net.P.cpu().numpy())
    assert loss == tr.loss_value()
