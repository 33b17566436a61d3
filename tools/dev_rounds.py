"""Conv time vs batch (number of workgroup rounds) for a short-K stem layer: separates per-launch fixed cost from per-round cost."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from facenet_amd import _lib
from tests.util import conv_desc, ptr
lib = _lib.load()


def bench(N, H, W, Cin, Cout, kh, kw, s, ph, pw, tile, dt=_lib.FN_F16, reps=40):
    tdt = torch.float16
    x = torch.randn(N, H, W, Cin, device='cuda').to(tdt)
    w = (torch.randn(Cout, kh, kw, Cin, device='cuda') * 0.05).to(tdt)
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
    d.tile_fwd = tile
    y = torch.zeros(N, d.OH, d.OW, Cout, dtype=tdt, device='cuda')
    b = torch.zeros(Cout, device='cuda')
    d.x, d.w, d.y, d.bias, d.relu = ptr(x), ptr(w), ptr(y), ptr(b), 1
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
    M = N * d.OH * d.OW
    bm, bn = tile // 1000, tile % 1000
    blocks = -(-M // bm) * -(-Cout // bn)
    print(f"N{N:4d} M={M:8d} blocks={blocks:6d} ({blocks / 256:6.1f}/CU) tile {bm}x{bn}: {us:8.2f} us  {us / max(blocks / 256, 1):6.2f} us per block-per-CU", flush=True)


for tile in (128064, 64064):
    print("2b 77x77x32->64 3x3")
    for N in (1, 2, 4, 8, 17, 34, 68, 136, 180):
        bench(N, 77, 77, 32, 64, 3, 3, 1, 0, 0, tile)
print("4a 37x37x80->192 3x3, 128x64")
for N in (2, 8, 34, 136, 180):
    bench(N, 37, 37, 80, 192, 3, 3, 1, 0, 0, 128064)
print("per-tile fixed vs per-k-tile cost, 128x64 tiles, N=180 77x77xCin->64 3x3")
for cin in (32, 64, 128, 256):
    bench(180, 77, 77, cin, 64, 3, 3, 1, 0, 0, 128064)
print("same, 64x64 tiles")
for cin in (32, 64, 128, 256):
    bench(180, 77, 77, cin, 64, 3, 3, 1, 0, 0, 64064)
