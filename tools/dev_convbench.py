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
    st = torch.zeros(2 * Cout, device='cuda')
    d.x, d.w, d.y = ptr(x), ptr(w), ptr(y)
    if stats: d.stats, d.stats_sq_off = ptr(st), Cout
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
print("--- block17-like M=5760 N=128, vary K")
for cin in (64, 128, 256, 512, 896, 1792): bench(90, 8, 8, cin, 128, 1, 1, 1, 0, 0)
print("--- M=11520 (N180)")
for cin in (64, 896): bench(180, 8, 8, cin, 128, 1, 1, 1, 0, 0)
print("--- block8-like M=810 N=192")
for cin in (64, 192, 576, 1792): bench(90, 3, 3, cin, 192, 1, 1, 1, 0, 0)
print("--- block35-like M=26010 N=32")
for cin in (32, 64, 256): bench(90, 17, 17, cin, 32, 1, 1, 1, 0, 0)
bench(90, 17, 17, 32, 32, 3, 3, 1, 1, 1)
print("--- big: 4a, 4b")
bench(90, 37, 37, 80, 192, 3, 3, 1, 0, 0); bench(90, 35, 35, 192, 256, 3, 3, 2, 0, 0); bench(90, 77, 77, 32, 64, 3, 3, 1, 0, 0)
print("--- stats epilogue on/off (bf16)")
bench(90, 8, 8, 896, 128, 1, 1, 1, 0, 0, _lib.FN_BF16, False); bench(90, 8, 8, 896, 128, 1, 1, 1, 0, 0, _lib.FN_BF16, True)
bench(90, 79, 79, 32, 32, 3, 3, 1, 0, 0, _lib.FN_BF16, False); bench(90, 79, 79, 32, 32, 3, 3, 1, 0, 0, _lib.FN_BF16, True)
