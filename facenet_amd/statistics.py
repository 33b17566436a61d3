"""``pairwise_similarities`` with the reference's signature and error behaviour (facenet/statistics.py:22-57),
computed by the wavefront-reduced fn_pairwise_sqdist kernel."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .engine import _ptr


def _decode_ord(i: int) -> float:
    i = int(i)
    bits = i if i >= 0 else (i ^ 0x7FFFFFFF)
    return float(np.array([bits & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])


def pairwise_similarities(xa, xb=None, metric: int = 0, atol: float = 1.e-5, device: str = "cuda"):
    """xa [n,E], xb [m,E] unit-norm rows -> 2(1 - xa.xb^T) (metric 0) or arccos (metric 1); with ``xb=None`` the strict
    upper triangle of xa against itself, flattened row-major (np.triu_indices order)."""
    lib = _lib.load()
    if metric not in (0, 1):
        raise ValueError("Undefined similarity metric {}".format(metric))      # statistics.py:55
    a = torch.as_tensor(np.asarray(xa) if not torch.is_tensor(xa) else xa).to(device=device, dtype=torch.float32).contiguous()
    b = a if xb is None else torch.as_tensor(np.asarray(xb) if not torch.is_tensor(xb) else xb).to(device=device, dtype=torch.float32).contiguous()
    n, m, E = a.shape[0], b.shape[0], a.shape[1]
    if n == 0 or m == 0:
        return np.zeros((0,) if xb is None else (n, m), dtype=np.float32)
    out = torch.empty(n, m, dtype=torch.float32, device=a.device)
    rng = torch.zeros(2, dtype=torch.int32, device=a.device)
    st = torch.cuda.current_stream(a.device).cuda_stream
    _lib.check(lib.fn_pairwise_sqdist(_ptr(a), _ptr(b), _ptr(out), _ptr(rng), n, m, E, metric, st), "pairwise_sqdist")
    if xb is None:
        iu = torch.triu_indices(n, n, offset=1, device=a.device)
        sims = out[iu[0], iu[1]]
        if sims.numel() == 0:
            return sims.cpu().numpy()
    else:
        sims = out
    lo, hi = (_decode_ord(v) for v in rng.cpu().tolist())
    lim = 1 + atol
    if lo < -lim or hi > lim:   # statistics.py:40-42 (the kernel reports min/max over the full matrix)
        raise ValueError("\nembeddings must be normalized to 1, range {} {}".format(lo, hi))
    return sims.cpu().numpy()
