"""Developer aid: what cold WEIGHTS cost a latency-bound convolution launch.  A Block17-sized layer (90 images, 8x8, 128 -> 128,
3x3: 102 workgroups of 64x64, 18 k tiles) replayed with R distinct weight tensors round-robin, activations shared:
R = 4: weights stay in L2; R = 1200 (350 MB): every launch streams its weights from HBM, as every layer of the real step does."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FN_CONV_HALO", "0")
from facenet_amd import _lib
from tests.util import conv_desc, ptr
lib = _lib.load()
N, H, W, Cin, Cout = 90, 8, 8, 128, 128
for kh, name in ((3, "3x3 K=1152"), (1, "1x1 K=128")):
  for tile in (64064, 64032):
    for R in (4, 1200):
        x = torch.randn(N, H, W, Cin, device='cuda').half()
        b = torch.zeros(Cout, device='cuda')
        ds, keep = [], []
        for r in range(R):
            d = conv_desc(N, H, W, Cin, Cout, kh, kh, 1, 0, 0, _lib.FN_F16)
            w = (torch.randn(Cout, kh, kh, Cin, device='cuda') * 0.05).half()
            if r == 0: y = torch.zeros(N, d.OH, d.OW, Cout, dtype=torch.float16, device='cuda')
            d.x, d.w, d.y, d.bias, d.relu, d.tile_fwd = ptr(x), ptr(w), ptr(y), ptr(b), 1, tile
            ds.append(d); keep.append(w)
        cur = torch.cuda.current_stream().cuda_stream
        for d in ds[:8]: _lib.check(lib.fn_conv2d_fwd(C.byref(d), cur))
        torch.cuda.synchronize()
        L = 240
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s_ = torch.cuda.current_stream().cuda_stream
            for i in range(L): lib.fn_conv2d_fwd(C.byref(ds[i % R]), s_)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(e) * 1e3 / L)
        print(f"{name} tile {tile // 1000}x{tile % 1000} R={R:5d}: {best:6.2f} us per launch", flush=True)
        del ds, keep, g
