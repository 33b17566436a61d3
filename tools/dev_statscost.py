"""Cost of the BN-statistics atomics in the conv epilogue: same launch with and without `stats` (graph replay x40)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from facenet_amd import _lib
from tests.util import conv_desc, ptr
lib = _lib.load()


def bench(N, H, W, Cin, Cout, kh, kw, s, ph, pw, stats, reps_r=16, reps=40):
    dt, tdt = _lib.FN_BF16, torch.bfloat16
    x = torch.randn(N, H, W, Cin, device='cuda').to(tdt)
    w = (torch.randn(Cout, kh, kw, Cin, device='cuda') * 0.05).to(tdt)
    d = conv_desc(N, H, W, Cin, Cout, kh, kw, s, ph, pw, dt)
    y = torch.zeros(N, d.OH, d.OW, Cout, dtype=tdt, device='cuda')
    st = torch.zeros(reps_r, 2 * Cout, dtype=torch.int64, device='cuda')
    d.x, d.w, d.y = ptr(x), ptr(w), ptr(y)
    if stats:
        d.stats, d.stats_sq_off, d.stats_replicas, d.stats_rep_stride = ptr(st), Cout, reps_r, 2 * Cout
    cur = torch.cuda.current_stream().cuda_stream
    lib.fn_conv2d_fwd(C.byref(d), cur); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s_ = torch.cuda.current_stream().cuda_stream
        for _ in range(reps): lib.fn_conv2d_fwd(C.byref(d), s_)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for name, sh in (("b35 3x3", (90, 17, 17, 32, 32, 3, 3, 1, 1, 1)), ("b35 1x1", (90, 17, 17, 256, 32, 1, 1, 1, 0, 0)), ("b17 1x7", (90, 8, 8, 128, 128, 1, 7, 1, 0, 3)),
                 ("b17 1x1", (90, 8, 8, 896, 128, 1, 1, 1, 0, 0)), ("4a", (90, 37, 37, 80, 192, 3, 3, 1, 0, 0)), ("2b", (90, 77, 77, 32, 64, 3, 3, 1, 0, 0)),
                 ("b8 1x3", (90, 3, 3, 192, 192, 1, 3, 1, 0, 1))):
    t0 = bench(*sh, stats=False)
    ts = {r: bench(*sh, stats=True, reps_r=r) for r in (1, 4, 16, 32, 64, 128)}
    print(f"{name:8s} no stats {t0:7.2f} us  " + "  ".join(f"R={r}: {t:6.2f}" for r, t in ts.items()), flush=True)
