"""Microbenchmark: what does normalise-on-load (+ the side write of the activated tensor) cost a small-map forward convolution?
Graph replay of 40 identical launches; modes: plain operand / NORM with 1, 4, 16 statistic replicas / NORM + nrm_z."""
import ctypes as C, sys, os, torch
sys.path.insert(0, '.')
from facenet_amd import _lib
from tests.util import conv_desc, ptr
lib = _lib.load()

def bench(name, N, H, W, Cin, Cout, kh, kw, s, ph, pw, mode, reps_stat=4, dt=_lib.FN_BF16, reps=40):
    tdt = torch.bfloat16
    x = torch.randn(N, H, W, Cin, device='cuda').to(tdt)
    z = torch.zeros_like(x)
    w = (torch.randn(Cout, kh, kw, Cin, device='cuda') * 0.05).to(tdt)
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
    y = torch.zeros(N, d.OH, d.OW, Cout, dtype=tdt, device='cuda')
    CB = 2048
    st = torch.zeros(16 * 2 * CB, device='cuda')
    stin = torch.rand(16 * 2 * CB, device='cuda') * 100
    beta = torch.zeros(CB, device='cuda')
    d.x, d.w, d.y = ptr(x), ptr(w), ptr(y)
    d.stats, d.stats_sq_off, d.stats_replicas, d.stats_rep_stride = ptr(st), CB, 4, 2 * CB
    if mode != "plain":
        d.nrm_stats, d.nrm_beta, d.nrm_sq_off, d.nrm_replicas, d.nrm_rep_stride = ptr(stin), ptr(beta), CB, reps_stat, 2 * CB
        d.nrm_count, d.nrm_eps = N * H * W, 1e-3
        if mode == "normz":
            d.nrm_z = ptr(z)
    cur = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.fn_conv2d_fwd(C.byref(d), cur)); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s_ = torch.cuda.current_stream().cuda_stream
        for _ in range(reps): lib.fn_conv2d_fwd(C.byref(d), s_)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    v = lib.fn_conv2d_variant(C.byref(d), 0)
    return best, v

shapes = [("b17 1x7", (90, 8, 8, 128, 128, 1, 7, 1, 0, 3)), ("b17 up", (90, 8, 8, 256, 896, 1, 1, 1, 0, 0)),
          ("b35 3x3", (90, 17, 17, 32, 32, 3, 3, 1, 1, 1)), ("b35 up", (90, 17, 17, 96, 256, 1, 1, 1, 0, 0)),
          ("b8 1x3", (90, 3, 3, 192, 192, 1, 3, 1, 0, 1)), ("b8 up", (90, 3, 3, 384, 1792, 1, 1, 1, 0, 0)),
          ("redA 3x3", (90, 17, 17, 192, 192, 3, 3, 1, 1, 1))]
for name, sh in shapes:
    row = []
    for mode, rs in (("plain", 0), ("norm", 1), ("norm", 4), ("norm", 16), ("normz", 4), ("normz", 16)):
        us, v = bench(name, *sh, mode=mode, reps_stat=max(rs, 1))
        row.append(f"{mode}{rs or ''}: {us:6.2f}")
    print(f"{name:9s} tile {v % 1000000 // 1000}x{v % 1000}  " + "  ".join(row))
