"""Data-parallel gradient exchange: the MI355X restatement of ``tf.distribute.MirroredStrategy()``
(apps/train_softmax_tf2_gpus.py:49-108): synchronous replicas, per-replica BatchNorm, gradients summed
across replicas, the same optimiser step everywhere.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The flat fp32 gradient buffer is cut
into buckets in BACKWARD order; a bucket is all-reduced on a communication stream as soon as the last weight
gradient inside it has been issued, overlapping the rest of backward.  The 1/world scale is applied inside the
fused optimiser (hyper[3]), so the collective is a plain SUM.

This module is device-agnostic on purpose: the bucketing logic and the exchange are covered by world-size-2
``gloo`` tests on CPU (tests/test_parallel.py)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

Bucket = Tuple[int, int, int]   # (ready_after_op_index, lo, hi) over the flat gradient buffer


def make_buckets(layer_offsets: Sequence[int], layer_sizes: Sequence[int], done_at: Dict[int, int], n_kernel: int,
                 tail: Tuple[int, int], n_bwd_ops: int, n_buckets: int = 6) -> List[Bucket]:
    """layer i owns [layer_offsets[i], +layer_sizes[i]) (declaration order = ascending offsets); ``done_at[i]`` is the
    index of the backward launch after which its gradient is complete.  Buckets are contiguous, walk the layers from
    last to first (the order backward finishes them), hold ~n_kernel/n_buckets elements each, and become ready in
    non-decreasing launch order.  ``tail`` = the [lo, hi) range of betas/biases, exchanged after the whole backward."""
    target = n_kernel / max(1, n_buckets)
    buckets: List[Bucket] = []
    hi, acc, ready = n_kernel, 0, 0
    for i in range(len(layer_offsets) - 1, -1, -1):
        acc += layer_sizes[i]
        ready = max(ready, done_at.get(i, 0))
        if acc >= target or i == 0:
            buckets.append((ready, layer_offsets[i], hi))
            hi, acc = layer_offsets[i], 0
    out: List[Bucket] = []
    r = 0
    for (rd, lo, h) in buckets:          # a later bucket may not be issued before an earlier one
        r = max(r, rd)
        out.append((r, lo, h))
    if tail[1] > tail[0]:
        out.append((n_bwd_ops, tail[0], tail[1]))
    return out


def check_buckets(buckets: Sequence[Bucket], n_kernel: int, tail: Tuple[int, int]) -> None:
    """Every gradient element is exchanged exactly once; readiness is monotone."""
    covered = sorted((lo, hi) for (_, lo, hi) in buckets)
    pos = 0
    for lo, hi in covered:
        if lo < pos:
            raise AssertionError(f"bucket [{lo},{hi}) overlaps the previous one")
        if lo > pos and not (pos == n_kernel and lo == tail[0]):
            raise AssertionError(f"gap [{pos},{lo}) is never all-reduced")
        pos = hi
    if pos != max(n_kernel, tail[1]):
        raise AssertionError("buckets do not reach the end of the gradient buffer")
    rd = [b[0] for b in buckets]
    if rd != sorted(rd):
        raise AssertionError("bucket readiness must be non-decreasing")


def allreduce_bucket(flat: torch.Tensor, lo: int, hi: int, group=None):
    """SUM all-reduce of flat[lo:hi] (a view: reduced in place)."""
    import torch.distributed as dist
    dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group)


def broadcast_parameters(tensors: Sequence[torch.Tensor], src: int = 0, group=None):
    """MirroredStrategy initialises every replica with the same variables; so do we (rank 0 wins)."""
    import torch.distributed as dist
    for t in tensors:
        dist.broadcast(t, src=src, group=group)
