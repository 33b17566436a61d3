"""Developer aid: run any script of this repo against an experimental build of the library:
    python tools/dev_withlib.py build/dbg/libX.so bench.py --steps 30 ..."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from facenet_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")
