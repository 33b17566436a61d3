"""Tile choice for memory-bound 1x1 layers (short K): run once per FN_CONV_TILE value."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from facenet_amd import _lib
from tests.util import conv_desc, ptr
lib = _lib.load()


def bench(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt, resid, reps=40):
    tdt = torch.bfloat16 if dt == _lib.FN_BF16 else torch.float16
    x = torch.randn(N, H, W, Cin, device='cuda').to(tdt)
    w = (torch.randn(Cout, kh, kw, Cin, device='cuda') * 0.05).to(tdt)
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
    y = torch.zeros(N, d.OH, d.OW, Cout, dtype=tdt, device='cuda')
    r = torch.randn(N, d.OH, d.OW, Cout, device='cuda').to(tdt)
    b = torch.zeros(Cout, device='cuda')
    st = torch.zeros(2 * Cout, device='cuda')
    d.x, d.w, d.y = ptr(x), ptr(w), ptr(y)
    if resid:
        d.resid, d.ld_res, d.scale, d.relu, d.bias = ptr(r), Cout, 0.17, 1, ptr(b)
    else:
        d.stats, d.stats_sq_off = ptr(st), Cout
    cur = torch.cuda.current_stream().cuda_stream
    lib.fn_conv2d_fwd(C.byref(d), cur); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s_ = torch.cuda.current_stream().cuda_stream
        for _ in range(reps): lib.fn_conv2d_fwd(C.byref(d), s_)
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1e3 / reps
    byts = 2.0 * N * (H * W * Cin + d.OH * d.OW * Cout * (2 if resid else 1))
    print(f"{os.environ.get('FN_CONV_TILE', 'auto'):>8s} N{N:3d} {H}x{W}x{Cin:4d}->{Cout:4d} k{kh}x{kw} resid={int(resid)} : {us:7.2f} us  {byts / us / 1e6:6.2f} TB/s", flush=True)


for N, dt in ((180, _lib.FN_F16), (90, _lib.FN_BF16)):
    bench(N, 17, 17, 96, 256, 1, 1, 1, 0, 0, dt, True)      # block35 up
    bench(N, 17, 17, 256, 32, 1, 1, 1, 0, 0, dt, False)     # block35 1x1
    bench(N, 8, 8, 256, 896, 1, 1, 1, 0, 0, dt, True)       # block17 up
    bench(N, 3, 3, 384, 1792, 1, 1, 1, 0, 0, dt, True)      # block8 up
    bench(N, 37, 37, 64, 80, 1, 1, 1, 0, 0, dt, False)      # 3b
