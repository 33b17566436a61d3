"""Loader throughput on the GPU box: synthetic 250x250 JPEG files (the VGGFace2/CASIA crops the reference trains on are
about that size) -> thread-pool decode -> pinned staging -> H2D -> fn_crop_or_pad_u8.  Prints images/s for several worker
counts, and the device-only rate of the crop kernel."""
import sys, tempfile, time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from facenet_amd import dataset
from facenet_amd.config import Config


def main():
    from PIL import Image
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        root = Path(d)
        for c in range(40):
            (root / f"c{c:02d}").mkdir()
            for i in range(25):
                base = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
                img = np.asarray(Image.fromarray(base).resize((250, 250), Image.BILINEAR))
                Image.fromarray(img).save(root / f"c{c:02d}" / f"{i:03d}.jpg", quality=90)
        db = dataset.Database(Config({"path": str(root)}))
        loader = dataset.ImageLoader(Config({"size": 160}))
        for workers, procs in ((1, False), (4, False), (4, True), (8, True), (16, True)):
            pipe = db.tf_dataset_api(loader, batch_size=100, buffer_size=10, repeat=True, workers=workers, prefetch=4, processes=procs)
            n, t0 = 0, None
            for images, labels in pipe:
                if t0 is None:
                    t0 = time.perf_counter()          # first batch = warm-up
                    continue
                n += images.shape[0]
                if n >= 6000:
                    break
            torch.cuda.synchronize()
            pipe.close()
            print(f"{'processes' if procs else 'threads  '} {workers:2d}: {n / (time.perf_counter() - t0):9.0f} images/s (decode + pack + H2D + crop)", flush=True)
        arrays = [loader.decode(f) for f in db.files[:100]]
        out = dataset.crop_or_pad_batch(arrays, 160)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = dataset.crop_or_pad_batch(arrays, 160, staging=getattr(out, "_staging", None))
        torch.cuda.synchronize()
        print(f"pack + H2D + crop only: {2000 / (time.perf_counter() - t0):9.0f} images/s", flush=True)


if __name__ == "__main__":
    main()
