"""Developer aid: the floor of a dependent launch inside a replayed HIP graph -- a chain of 400 launches of the smallest kernel of the
library (fn_acc_to_float on one element), time per launch."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facenet_amd import _lib
from tests.util import ptr
lib = _lib.load()
src = torch.zeros(1024, dtype=torch.int64, device='cuda')
dst = torch.zeros(1024, dtype=torch.float32, device='cuda')
for n in (1, 1024):
    for L in (400,):
        cur = torch.cuda.current_stream().cuda_stream
        lib.fn_acc_to_float(ptr(src), ptr(dst), n, 40, cur)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s_ = torch.cuda.current_stream().cuda_stream
            for _ in range(L): lib.fn_acc_to_float(ptr(src), ptr(dst), n, 40, s_)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(e) * 1e3 / L)
        print(f"{L} dependent launches of a {n}-element kernel: {best:.2f} us per launch", flush=True)
