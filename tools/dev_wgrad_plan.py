"""Developer aid (host only, no GPU): how the grouped weight gradients of the batch-90 training plan are planned --
per layer the kernel variant, workgroups, pixel splits and slab bytes (fn_conv2d_wgrad_group_build sizing call)."""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facenet_amd import _lib
from facenet_amd.engine import Network, Lowering

N = int(sys.argv[1]) if len(sys.argv) > 1 else 90
net = Network(embedding_size=128, allocate=False)
g = Lowering(net, N=N, training=True, declare=True)
net._topology(g)
lib = _lib.load()
descs = []
for r in g.recs:
    if r.kind != "conv":
        continue
    L = r.layer
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.Cin = N, r.x.buf.H, r.x.buf.W, L.cin
    d.OH, d.OW, d.Cout = r.y.buf.H, r.y.buf.W, L.cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w = L.kh, L.kw, L.stride, L.pad_h, L.pad_w
    d.dtype, d.ld_x, d.ld_y, d.scale = 0, r.x.buf.C, r.y.buf.C, 1.0
    d.x = d.y = d.dw = 4096
    descs.append((L.name, d))
groups = collections.defaultdict(list)
for name, d in descs:
    groups[lib.fn_conv2d_variant(C.byref(d), 2)].append((name, d))
nb = lib.fn_conv2d_wgrad_arg_bytes()
tot_ws = 0
for v, mem in sorted(groups.items()):
    n = len(mem)
    arr = (_lib.ConvDesc * n)(*[m[1] for m in mem])
    ha = (C.c_uint8 * (nb * n))(); hp = (C.c_int32 * (n + 1))(); ws = C.c_int64(0)
    total = lib.fn_conv2d_wgrad_group_build(arr, n, v, ha, hp, None, C.byref(ws))
    if total < 0:
        print(v, "error", lib.fn_last_error()); continue
    print(f"variant {v}: {n} layers, {total} workgroups, slabs {ws.value * 4 / 1e6:.1f} MB")
    tot_ws += ws.value
    if "-v" in sys.argv:
        for i, (name, d) in enumerate(mem):
            one = (_lib.ConvDesc * 1)(d); h1 = (C.c_uint8 * nb)(); p1 = (C.c_int32 * 2)(); w1 = C.c_int64(0)
            t1 = lib.fn_conv2d_wgrad_group_build(one, 1, v, h1, p1, None, C.byref(w1))
            numel = d.Cout * d.KH * d.KW * d.Cin
            print(f"   {name:50s} {d.H}x{d.W}x{d.Cin}->{d.Cout} k{d.KH}x{d.KW}s{d.stride}  wgs {t1:5d}  splits {w1.value // numel if w1.value else 1:4d}  slab {w1.value * 4 / 1e6:7.2f} MB")
print(f"total slabs {tot_ws * 4 / 1e6:.1f} MB; dW itself {sum(d.Cout * d.KH * d.KW * d.Cin for _, d in descs) * 4 / 1e6:.1f} MB")
