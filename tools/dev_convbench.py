"""Microbenchmark of fn_conv2d_fwd: fixed cost vs per-K-tile cost (graph replay of 40 identical launches)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from facenet_amd import _lib
from tests.util import conv_desc, ptr
lib = _lib.load()
def bench(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt=_lib.FN_F16, stats=False, reps=40):
    tdt = torch.float16 if dt == _lib.FN_F16 else torch.bfloat16
    x = torch.randn(N, H, W, Cin, device='cuda').to(tdt)
    w = (torch.randn(Cout, kh, kw, Cin, device='cuda') * 0.05).to(tdt)
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
    y = torch.zeros(N, d.OH, d.OW, Cout, dtype=tdt, device='cuda')
    st = torch.zeros(16 * 2 * Cout, dtype=torch.int64, device='cuda')
    d.x, d.w, d.y = ptr(x), ptr(w), ptr(y)
    if stats: d.stats, d.stats_sq_off, d.stats_replicas, d.stats_rep_stride = ptr(st), Cout, 16, 2 * Cout
    cur = torch.cuda.current_stream().cuda_stream
    lib.fn_conv2d_fwd(C.byref(d), cur); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s_ = torch.cuda.current_stream().cuda_stream
        for _ in range(reps): lib.fn_conv2d_fwd(C.byref(d), s_)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    fl = 2.0 * N * d.OH * d.OW * Cout * kh * kw * Cin
    v = lib.fn_conv2d_variant(C.byref(d), 0)
    print(f"N{N} {H}x{W}x{Cin}->{Cout} k{kh}x{kw}s{s} tile {v//1000}x{v%1000} K={kh*kw*Cin:5d} ktiles={(kh*kw*Cin+63)//64:3d} : {us:7.2f} us  {fl/us/1e6:7.1f} TF/s stats={stats}")
import os
shapes = [("b17 1x1", (90, 8, 8, 896, 128, 1, 1, 1, 0, 0)), ("b17 1x7", (90, 8, 8, 128, 128, 1, 7, 1, 0, 3)), ("b17 up", (90, 8, 8, 256, 896, 1, 1, 1, 0, 0)),
          ("b35 1x1", (90, 17, 17, 256, 32, 1, 1, 1, 0, 0)), ("b35 3x3", (90, 17, 17, 32, 32, 3, 3, 1, 1, 1)), ("b35 up", (90, 17, 17, 96, 256, 1, 1, 1, 0, 0)),
          ("b8 1x1", (90, 3, 3, 1792, 192, 1, 1, 1, 0, 0)), ("b8 1x3", (90, 3, 3, 192, 192, 1, 3, 1, 0, 1)), ("b8 up", (90, 3, 3, 384, 1792, 1, 1, 1, 0, 0)),
          ("redA 3x3", (90, 17, 17, 192, 192, 3, 3, 1, 1, 1)), ("4a", (90, 37, 37, 80, 192, 3, 3, 1, 0, 0)), ("4b", (90, 35, 35, 192, 256, 3, 3, 2, 0, 0)), ("2b", (90, 77, 77, 32, 64, 3, 3, 1, 0, 0))]
if len(sys.argv) > 1 and sys.argv[1] == "halo":
    big = [("2a", (90, 79, 79, 32, 32, 3, 3, 1, 0, 0)), ("2b", (90, 77, 77, 32, 64, 3, 3, 1, 0, 0)), ("4a", (90, 37, 37, 80, 192, 3, 3, 1, 0, 0)),
           ("2a x180", (180, 79, 79, 32, 32, 3, 3, 1, 0, 0)), ("2b x180", (180, 77, 77, 32, 64, 3, 3, 1, 0, 0)), ("4a x180", (180, 37, 37, 80, 192, 3, 3, 1, 0, 0))]
    for name, sh in big:
        for st_ in (True, False):
            print(f"{name:9s}", end=" "); bench(*sh, dt=_lib.FN_BF16, stats=st_)
    sys.exit(0)
for tile in (None, "32x32x4", "32x64x4", "64x32x4", "64x64x4", "64x64x1", "64x128x4", "64x128x1", "128x64x2", "128x128x2", "128x64x1", "128x128x1"):
    if tile: os.environ["FN_CONV_TILE"] = tile
    print("=== tile", tile or "auto")
    for name, sh in shapes:
        print(f"{name:9s}", end=" "); bench(*sh, dt=_lib.FN_BF16, stats=True)
