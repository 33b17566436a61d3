"""Developer aid: the stem 3x3 layers (180 images, f16) under every implicit-GEMM tile (tile_fwd set per descriptor; FN_CONV_HALO=0 keeps
the halo kernel out): does the time follow the operand bytes per tile (L2 -> CU bound) or the MFMA work?
    python tools/dev_stemtiles.py [layer [tile]]           e.g. 4a 128128
    FN_DEV_LIB=build/dbg/libN.so python tools/dev_stemtiles.py ...    an experimental build (e.g. -DFN_IG_DBG=7: what is left of the
                                                                       kernel without loads / multiply / LDS stores)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FN_CONV_HALO", "0")
from facenet_amd import _lib
from tests.util import conv_desc, ptr
if os.environ.get('FN_DEV_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['FN_DEV_LIB'])     # an experimental build of the library
lib = _lib.load()
R = 6


def bench(name, N, H, W, Cin, Cout, s, tile, dt=_lib.FN_F16):
    tdt = torch.float16
    ds, keep = [], []
    for r in range(R):
        d = conv_desc(N, H, W, Cin, Cout, 3, 3, s, 0, 0, dt)
        x = torch.randn(N, H, W, Cin, device='cuda').to(tdt)
        w = (torch.randn(Cout, 3, 3, Cin, device='cuda') * 0.05).to(tdt)
        y = torch.zeros(N, d.OH, d.OW, Cout, dtype=tdt, device='cuda')
        b = torch.zeros(Cout, device='cuda')
        d.x, d.w, d.y, d.bias, d.relu, d.tile_fwd = ptr(x), ptr(w), ptr(y), ptr(b), 1, tile
        ds.append(d); keep.append((x, w, y, b))
    cur = torch.cuda.current_stream().cuda_stream
    for d in ds: _lib.check(lib.fn_conv2d_fwd(C.byref(d), cur))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s_ = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            for d in ds: lib.fn_conv2d_fwd(C.byref(d), s_)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) * 1e3 / (3 * R))
    fl = 2.0 * N * ds[0].OH * ds[0].OW * Cout * 9 * Cin
    bm, bn = (tile // 1000, tile % 1000) if tile < 9000000 else (128, 64)      # 9000000: the halo-tile kernel (8 or 16 x 16 pixels)
    print(f"{name:10s} tile {bm:3d}x{bn:3d}: {best:7.2f} us {fl / best / 1e6:7.1f} TF/s   operand bytes per MFMA clock {(bm + bn) * 128 / (bm * bn / 32):5.1f}", flush=True)


only_layer = sys.argv[1] if len(sys.argv) > 1 else None          # python tools/dev_stemtiles.py [layer [tile]]
only_tile = int(sys.argv[2]) if len(sys.argv) > 2 else None
for name, shp in (("b17", (90, 8, 8, 128, 128, 1)), ("2b", (180, 77, 77, 32, 64, 1)), ("4a", (180, 37, 37, 80, 192, 1)), ("4b", (180, 35, 35, 192, 256, 2)), ("redA0b", (180, 19, 19, 192, 192, 1))):
    if only_layer and name != only_layer: continue
    for tile in (128128, 128064, 128032, 64128, 64064, 32032, 9000000):
        if only_tile and tile != only_tile: continue
        if tile % 1000 > 64 and shp[4] <= 64: continue
        bench(name, *shp, tile)
