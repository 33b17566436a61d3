"""Triplet selection and loss (build-defined: the reference has none; arXiv 1503.03832 sec. 3).
Thin host wrappers over fn_pairwise_sqdist / fn_select_triplets / fn_triplet_loss_fwd_bwd for use outside a plan."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .engine import _ptr


def squared_distances(emb: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    e = emb.to(dtype=torch.float32).contiguous()
    n, E = e.shape
    out = torch.empty(n, n, dtype=torch.float32, device=e.device)
    _lib.check(lib.fn_pairwise_sqdist(_ptr(e), _ptr(e), _ptr(out), None, n, n, E, 2, torch.cuda.current_stream(e.device).cuda_stream),
               "pairwise_sqdist")
    return out


def select_triplets(dist: torch.Tensor, labels, alpha: float, nrof_triplets: int, seed: int = 0, semi_hard: bool = False):
    """Returns (triplets int32 [T,3] on device, info dict).  Raises ValueError when the pool has too few positive pairs."""
    lib = _lib.load()
    n = dist.shape[0]
    lab = torch.as_tensor(np.asarray(labels), dtype=torch.int32).to(dist.device)
    trip = torch.zeros(nrof_triplets, 3, dtype=torch.int32, device=dist.device)
    info = torch.zeros(8 + 5 * (n * (n - 1) // 2), dtype=torch.int32, device=dist.device)
    _lib.check(lib.fn_select_triplets(_ptr(dist.contiguous()), _ptr(lab), n, float(alpha), nrof_triplets, int(seed) & 0xFFFFFFFF,
                                      1 if semi_hard else 0, _ptr(trip), _ptr(info), torch.cuda.current_stream(dist.device).cuda_stream),
               "select_triplets")
    q, valid, short = info[:3].cpu().tolist()
    if short:
        raise ValueError("pool too small for the requested number of triplets")
    return trip, {"pairs": q, "valid": valid}


def triplet_loss(emb: torch.Tensor, alpha: float, with_grad: bool = False):
    """emb fp32 [3T,E] rows (a0,p0,n0,...).  Returns loss (0-d tensor) and, optionally, d loss / d emb."""
    lib = _lib.load()
    e = emb.to(dtype=torch.float32).contiguous()
    T, E = e.shape[0] // 3, e.shape[1]
    loss = torch.zeros(4, dtype=torch.float32, device=e.device)      # word 0 = the loss, words 1-3 = the launch's accumulator
    grad = torch.empty_like(e) if with_grad else None
    _lib.check(lib.fn_triplet_loss_fwd_bwd(_ptr(e), _ptr(grad) if with_grad else None, _ptr(loss), T, E, float(alpha),
                                           torch.cuda.current_stream(e.device).cuda_stream), "triplet_loss")
    return (loss[0], grad) if with_grad else loss[0]
