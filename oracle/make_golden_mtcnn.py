#!/usr/bin/env python3
"""Generates tests/golden/mtcnn_oracle_run.npz from oracle/mtcnn_oracle.py: one seeded image, synthetic weights, the cascade's
stage outputs.  SELF-GENERATED (parity unpinned, see the oracle's header): it pins the restatement against drift and gives the
GPU tests a committed expected output.
    python oracle/make_golden_mtcnn.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mtcnn_oracle as mo  # noqa: E402

SEED, FACE_BIAS = 7, (0.5, 1.0, 1.0)


def main():
    rng = np.random.default_rng(SEED)
    base = rng.integers(0, 256, (14, 18, 3), dtype=np.uint8)
    img = np.clip(np.kron(base, np.ones((8, 8, 1), np.uint8)).astype(np.int32) + rng.integers(-12, 13, (112, 144, 3)), 0, 255).astype(np.uint8)
    tr = {}
    faces = mo.detect_faces(img, mo.Nets(mo.random_weights(SEED, face_bias=FACE_BIAS)), trace=tr)
    out = os.path.join(ROOT, "tests", "golden", "mtcnn_oracle_run.npz")
    np.savez_compressed(out, image=img, weights_seed=SEED, face_bias=np.asarray(FACE_BIAS), stage1=tr["stage1"], stage2=tr["stage2"],
                        stage3=tr["stage3"], points=tr["points"])
    print(out, {k: v.shape for k, v in tr.items()}, len(faces), "faces")


if __name__ == "__main__":
    main()
