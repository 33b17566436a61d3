"""Developer aid: where a workgroup of the implicit-GEMM kernel spends its life, inside the real step.  Needs a debug build of the
library with per-workgroup phase clocks (not the shipped one):

    cd facenet_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DFN_IG_DBG=32 -c conv_igemm.hip -o ../../build/dbg/ig32.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC ../../build/dbg/ig32.o $(ls ../../build/obj/*.o | grep -v conv_igemm) -o ../../build/dbg/lib32.so
    python tools/dev_phases.py [bench.py arguments]

Prints, per (tile, epilogue path): workgroups, mean k tiles, and the mean time (us, 100 MHz wall clock) a workgroup spends in
index / tap-table prologue | first tile (load -> LDS -> barrier) | k loop | epilogue = C tile through LDS + row passes + folds and
atomics + acknowledgement of the last store.  With several workgroups per CU the phases of one overlap the others': the
numbers say what a workgroup waits for, not what the chip is busy with."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from facenet_amd import _lib
_lib.LIB_PATH = os.environ.get("FN_DEV_LIB") or os.path.join(ROOT, "build", "dbg", "lib32.so")
lib = _lib.load()
import bench

sys.argv = ["bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"] + sys.argv[1:]
buf = (C.c_ulonglong * (576 * 16))()
try:
    bench.main()
finally:
    lib.fn_debug_phases.restype = C.c_int
    assert lib.fn_debug_phases(buf, 0) == 0
    t = np.array(list(buf), dtype=np.float64).reshape(576, 16)
    names = {1: 32, 2: 64, 3: 128}
    kinds = ["plain", "resid", "bn-bwd", "res-bwd", "generic", "accum"]
    print(f"{'tile':>9s} {'epilogue':>12s} {'1x1':>4s} {'WGs':>9s} {'k tiles':>8s} | {'index':>6s} {'first':>6s} {'loop':>7s} {'per kt':>6s} {'epil':>6s} = {'C->LDS':>6s} {'rows':>6s} {'folds':>6s} {'ack':>6s}  us per workgroup | share with f32/prelu accumulate mask/out2 bn_y resid")
    for i in range(576):
        n = t[i, 0]
        if n == 0: continue
        mode = i % 3
        tile, kind = (i // 3) // 12, (i // 3) % 12
        bm, bn = names.get(tile // 4, 0), names.get(tile % 4, 0)
        u = t[i, 1:5] / n / 100.0
        kt = t[i, 5] / n
        e = t[i, 7:11] / n / 100.0
        label = kinds[kind % 6] + ("+st" if kind >= 6 else "") + ("", "/norm", "/sib")[mode]
        print(f"{bm:4d}x{bn:<4d} {label:>12s} {t[i, 6] / n:4.2f} {int(n):9d} {kt:8.1f} | {u[0]:6.2f} {u[1]:6.2f} {u[2]:7.2f} {u[2] / max(kt, 1):6.3f} {u[3]:6.2f} = {e[0]:6.2f} {e[1]:6.2f} {e[2]:6.2f} {e[3]:6.2f} | " + " ".join(f"{t[i, 11 + k] / n:4.2f}" for k in range(5)))
