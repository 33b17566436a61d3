"""Developer aid: does it cost a launch to be a DIFFERENT kernel than its predecessors (cold instruction cache)?  A Block17-sized 3x3 layer
(102 workgroups) launched 240 times in one graph: one tile variant throughout, or four variants (+ the 1x1 kernel on another layer) in rotation.
The rotation's mean is compared with the mean of the variants run alone."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FN_CONV_HALO", "0")
from facenet_amd import _lib
from tests.util import conv_desc, ptr
lib = _lib.load()
N, H, W, Cin, Cout = 90, 8, 8, 128, 128
x = torch.randn(N, H, W, Cin, device='cuda').half()
b = torch.zeros(Cout, device='cuda')
keep = []
def make(kh, tile):
    d = conv_desc(N, H, W, Cin, Cout, kh, kh, 1, 0, 0, _lib.FN_F16)
    w = (torch.randn(Cout, kh, kh, Cin, device='cuda') * 0.05).half()
    y = torch.zeros(N, d.OH, d.OW, Cout, dtype=torch.float16, device='cuda')
    d.x, d.w, d.y, d.bias, d.relu, d.tile_fwd = ptr(x), ptr(w), ptr(y), ptr(b), 1, tile
    keep.append((w, y)); return d
def run(ds, L=240):
    cur = torch.cuda.current_stream().cuda_stream
    for d in ds: _lib.check(lib.fn_conv2d_fwd(C.byref(d), cur))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s_ = torch.cuda.current_stream().cuda_stream
        for i in range(L): lib.fn_conv2d_fwd(C.byref(ds[i % len(ds)]), s_)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) * 1e3 / L)
    return best
variants = [(3, 64064), (3, 64032), (3, 32032), (3, 32064), (1, 64064), (1, 64032), (3, 128032), (3, 128064)]
alone = []
for kh, t in variants:
    us = run([make(kh, t) for _ in range(4)])
    alone.append(us)
    print(f"{kh}x{kh} tile {t // 1000}x{t % 1000} alone: {us:6.2f} us", flush=True)
rot = run([make(kh, t) for kh, t in variants])
print(f"rotation of the {len(variants)} kernels: {rot:6.2f} us per launch; mean of the kernels alone {sum(alone) / len(alone):6.2f} us")
