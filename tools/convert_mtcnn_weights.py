#!/usr/bin/env python3
"""Offline, explicitly opted-in converter: the PyPI `mtcnn` package's `mtcnn_weights.npy` (a pickled
{'pnet': [...], 'rnet': [...], 'onet': [...]} of Keras get_weights() lists) -> an .npz of named variables that
facenet_amd.detectors.mtcnn.load_weights accepts.

The product path never unpickles (a pickle executes code from the file).  This tool reads the pickle payload with a
RESTRICTED unpickler that can only rebuild numpy arrays, lists, tuples and dicts; any other global in the stream aborts.

    python tools/convert_mtcnn_weights.py --i-trust-this-file mtcnn_weights.npy mtcnn_weights.npz
"""
import argparse
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
            ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}


class _ArraysOnly(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to load global {module}.{name}: only numpy arrays, lists and dicts are accepted")


def read_pickled_npy(path):
    with open(path, "rb") as f:
        version = np.lib.format.read_magic(f)
        shape, _, dtype = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
        if not dtype.hasobject:
            raise ValueError(f"{path} holds a plain numeric array of shape {shape}, not the package's weight dictionary")
        obj = _ArraysOnly(f).load()
    return obj.item() if isinstance(obj, np.ndarray) and obj.shape == () else obj


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--i-trust-this-file", action="store_true", help="required: confirms the source of the pickle is trusted")
    a = ap.parse_args()
    if not a.i_trust_this_file:
        ap.error("pass --i-trust-this-file to convert a pickled weight file")
    if not a.dst.endswith(".npz"):
        ap.error("the destination must end in .npz")
    from facenet_amd.detectors.mtcnn import weights_from_lists
    lists = read_pickled_npy(a.src)
    if not isinstance(lists, dict) or not {"pnet", "rnet", "onet"} <= set(lists):
        raise SystemExit("unexpected content: want a dict with 'pnet', 'rnet', 'onet'")
    np.savez(a.dst, **weights_from_lists(lists))
    print(f"wrote {a.dst}")


if __name__ == "__main__":
    main()
