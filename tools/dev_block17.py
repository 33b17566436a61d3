"""Microbenchmark of fn_block17_infer: whole block and (FN_B17_STOP=k) its first k stages; graph replay of 20 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facenet_amd import _lib
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 180
dt = torch.float16
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(N, 8, 8, 896, device='cuda', generator=g).to(dt)
y = torch.zeros_like(x)
def w(*shape): return (torch.randn(*shape, device='cuda', generator=g) * 0.03).to(dt)
ws = [w(128, 896), w(128, 896), w(128, 7, 128), w(128, 7, 128), w(896, 256)]
bs = [torch.zeros(128, device='cuda') for _ in range(4)] + [torch.zeros(896, device='cuda')]
def run(st):
    return lib.fn_block17_infer(x.data_ptr(), y.data_ptr(), N, *[t.data_ptr() for t in ws], *[t.data_ptr() for t in bs], 0.1, 1, _lib.FN_F16, st)
cur = torch.cuda.current_stream().cuda_stream
_lib.check(run(cur)); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    s_ = torch.cuda.current_stream().cuda_stream
    for _ in range(20): run(s_)
gr.replay(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) * 1e3 / 20)
print(f"N={N} stop={os.environ.get('FN_B17_STOP', '0')}: {best:7.2f} us per launch")

