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
            tr.M.zero_(); tr.V.zero_(); tr.hyper.copy_(torch.tensor([0.01, 1.0, 1.0, 1.0 / world]))
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
    # single-process emulation: two replicas' backward passes, gradients averaged by hand, one optimiser step
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
        st = net.stream()
        for ops in (tr.pre_ops, tr.plan.fwd, tr.loss_ops, tr.plan.bwd):
            tr.plan.run_ops(ops, st)
        torch.cuda.synchronize()
        grads.append(tr.G.clone())
    p0 = net.P.clone()
    tr.G.copy_((grads[0] + grads[1]) / world)
    tr.plan.run_ops(tr.opt_ops, net.stream())
    torch.cuda.synchronize()
    upd_ref = (net.P - p0).cpu().numpy()
    upd_dp = res[0][1] - p0.cpu().numpy()
    gsum = (grads[0] + grads[1]).cpu().numpy()
    cos_g = float(np.dot(gsum, res[0][2]) / (np.linalg.norm(gsum) * np.linalg.norm(res[0][2])))
    cos_u = float(np.dot(upd_ref, upd_dp) / (np.linalg.norm(upd_ref) * np.linalg.norm(upd_dp)))
    print(f"graph={use_graph}: cosine(summed grads) {cos_g:.4f}, cosine(param update) {cos_u:.4f}, |update| {np.linalg.norm(upd_dp):.4f} vs {np.linalg.norm(upd_ref):.4f}")
    assert cos_g > 0.97 and cos_u > 0.97       # limited by run-to-run training noise (fp32 atomics + storage rounding)
    assert abs(np.linalg.norm(upd_dp) / np.linalg.norm(upd_ref) - 1) < 0.05
