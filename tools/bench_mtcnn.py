"""BASELINE.json config 5: MTCNN P/R/O-Net cascade on 1280x720 frames, one MI355X.  Synthetic weights (the package's trained
file is not available offline) with the P-Net face bias chosen so that stage 1 passes a realistic number of cells.
    python tools/bench_mtcnn.py [--frames 20] [--bias -0.3] [--cpu]   ->  one JSON line (ms per frame, stage split, CPU oracle)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from facenet_amd.detectors import mtcnn as gm
from oracle import mtcnn_oracle as mo   # weights generator + the CPU leg only


def frame(h, w, seed, cell=24):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (-(-h // cell), -(-w // cell), 3), dtype=np.uint8)
    img = np.kron(base, np.ones((cell, cell, 1), np.uint8))[:h, :w].astype(np.int32) + rng.integers(-12, 13, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--bias", type=float, default=-0.3)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    weights = mo.random_weights(0, face_bias=(a.bias, 1.0, 1.0))
    det = gm.MTCNN(weights=weights)
    frames = [torch.from_numpy(frame(720, 1280, s)).cuda() for s in range(4)]
    for f in frames:
        det.detect_boxes(f)
    torch.cuda.synchronize()
    split = {"stage1": 0.0, "stage2": 0.0, "stage3": 0.0}
    n1 = n2 = n3 = 0
    t0 = time.perf_counter()
    for i in range(a.frames):
        f = frames[i % len(frames)]
        t = time.perf_counter(); s1 = det._stage1(f); split["stage1"] += time.perf_counter() - t
        t = time.perf_counter(); s2 = det._stage2(f, s1); split["stage2"] += time.perf_counter() - t
        t = time.perf_counter(); s3, _ = det._stage3(f, s2); split["stage3"] += time.perf_counter() - t
        n1, n2, n3 = n1 + len(s1), n2 + len(s2), n3 + len(s3)
    dt = time.perf_counter() - t0
    # device time of stage 1 alone (pyramid + P-Net + compaction, 11 levels), events on the launch stream
    st = torch.cuda.current_stream().cuda_stream
    pyr = det._pyramid(720, 1280)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        for i in range(len(pyr["levels"])):
            det._run_level(frames[0], pyr, i, st, pyr["rec"], reset=(i == 0))
    e1.record(); torch.cuda.synchronize()
    out = {"metric": "MTCNN detect_faces, 1280x720 frames", "value": round(a.frames / dt, 2), "unit": "frames/sec", "ms_per_frame": round(1e3 * dt / a.frames, 2),
           "ms_split": {k: round(1e3 * v / a.frames, 2) for k, v in split.items()}, "stage1_device_ms": round(e0.elapsed_time(e1) / 10, 3),
           "boxes_per_frame": {"stage1": n1 / a.frames, "stage2": n2 / a.frames, "faces": n3 / a.frames}, "pnet_face_bias": a.bias,
           "data": "synthetic frames, synthetic weights", "dtype": "f16"}
    if a.cpu:
        img = frames[0].cpu().numpy()
        t = time.perf_counter()
        mo.detect_faces(img, mo.Nets(weights))
        out["cpu_oracle_ms_per_frame"] = round(1e3 * (time.perf_counter() - t), 1)
        out["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
