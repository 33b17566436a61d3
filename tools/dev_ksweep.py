"""Fixed cost vs per-k-tile cost of the small-tile convolution (graph replay of 40 identical launches, warm L2).
Run under FN_CONV_KS=1 / unset to compare the in-launch split-K."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from facenet_amd import _lib
from tests.util import conv_desc, ptr
lib = _lib.load()


def bench(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt=_lib.FN_BF16, reps=40):
    tdt = torch.bfloat16
    x = torch.randn(N, H, W, Cin, device='cuda').to(tdt)
    w = (torch.randn(Cout, kh, kw, Cin, device='cuda') * 0.05).to(tdt)
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
    y = torch.zeros(N, d.OH, d.OW, Cout, dtype=tdt, device='cuda')
    st = torch.zeros(16, 2 * Cout, dtype=torch.int64, device='cuda')      # 16 statistic replicas, as the engine uses for M >= 32K
    d.x, d.w, d.y, d.stats, d.stats_sq_off, d.stats_replicas, d.stats_rep_stride = ptr(x), ptr(w), ptr(y), ptr(st), Cout, 16, 2 * Cout
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
    v = lib.fn_conv2d_variant(C.byref(d), 0)
    M = N * d.OH * d.OW
    print(f"N{N:3d} {H}x{W}x{Cin:4d}->{Cout:4d} k{kh}x{kw} M={M:6d} variant {v:8d} ktiles={(kh*kw*Cin+63)//64:3d} : {us:7.2f} us", flush=True)


for N in (90, 180):
    for cin in (64, 128, 256, 512):        # 1x7: K = 7*cin
        bench(N, 8, 8, cin, 128, 1, 7, 1, 0, 3)
    for cin in (256, 896, 1792):           # 1x1
        bench(N, 8, 8, cin, 128, 1, 1, 1, 0, 0)
    bench(N, 3, 3, 1792, 192, 1, 1, 1, 0, 0)
    bench(N, 3, 3, 192, 192, 1, 3, 1, 0, 1)
    bench(N, 17, 17, 32, 32, 3, 3, 1, 1, 1)

print("--- floor: tiny kernels in a graph")
hyper = torch.tensor([0.01, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0], device='cuda')
def floor(fn, reps=40):
    g = torch.cuda.CUDAGraph()
    fn(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        s_ = torch.cuda.current_stream().cuda_stream
        for _ in range(reps): fn(s_)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
import inspect
print("adam_tick %.2f us" % floor(lambda s: lib.fn_adam_tick(ptr(hyper), 0.9, 0.999, s)))
z = torch.zeros(64, device='cuda')
print("N=1 1x1 K=8 conv (1 block):", end=" "); bench(1, 1, 1, 8, 32, 1, 1, 1, 0, 0)
print("N=1 1x1 K=896 conv (1x4 blocks):", end=" "); bench(1, 8, 8, 896, 128, 1, 1, 1, 0, 0)
print("N=1 1x7 K=896 conv:", end=" "); bench(1, 8, 8, 128, 128, 1, 7, 1, 0, 3)
print("N=8 1x7 K=896 conv:", end=" "); bench(8, 8, 8, 128, 128, 1, 7, 1, 0, 3)
print("N=30 1x7 K=896 conv:", end=" "); bench(30, 8, 8, 128, 128, 1, 7, 1, 0, 3)
