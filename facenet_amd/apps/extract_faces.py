# coding:utf-8
"""Face extraction entry point, same behaviour as the reference's apps/extract_faces.py:16-93 with the MTCNN detector on the
MI355X: every image of every class directory -> detector -> (optionally only single-face images) -> crop with margin ->
antialiased resize -> ``<outdir>/<class>/<stem>.png`` (further faces of one image: ``<stem>_<n>.png``).  The reference stores
the box sizes in an h5 file (h5py is not installed here): they go to ``sizes.json`` next to the thumbnails instead.
``python -m facenet_amd.apps.extract_faces --config x.yaml`` (keys: dataset.path, outdir, image.size, image.margin,
detect_multiple_faces, detector, mtcnn.weights_file)."""
from __future__ import annotations

import json
from pathlib import Path

import click
import numpy as np
from PIL import Image

from facenet_amd import config as config_mod
from facenet_amd import dataset
from facenet_amd.detectors.face_detector import FaceDetector, image_processing


def extract_faces(classes, outdir, detector, image_options, detect_multiple_faces: bool = False, log=print):
    """classes: iterable of objects with .name and .files.  Returns {'extracted': images that produced thumbnails,
    'unread': files PIL could not open, 'sizes': {relative png path: [box height, box width]}}."""
    outdir = Path(outdir)
    stats = {"extracted": 0, "unread": 0, "sizes": {}}
    for cls in classes:
        cls_dir = outdir.joinpath(cls.name)
        cls_dir.mkdir(parents=True, exist_ok=True)
        for path in cls.files:
            target = cls_dir.joinpath(Path(path).stem + '.png')
            try:
                img = Image.open(path).convert(detector.mode)
                pixels = np.asarray(img, dtype=np.uint8)
            except Exception:
                stats["unread"] += 1
                continue
            boxes = detector.detect(pixels)
            if len(boxes) == 0 or (len(boxes) > 1 and not detect_multiple_faces):
                continue
            stats["extracted"] += 1
            for n, box in enumerate(boxes):
                name = target if n == 0 else target.parent.joinpath('{}_{}{}'.format(target.stem, n, target.suffix))
                image_processing(img, box, image_options).save(name)
                stats["sizes"][str(name.relative_to(outdir))] = [int(box.height), int(box.width)]
    with outdir.joinpath("sizes.json").open("w") as fh:
        json.dump(stats["sizes"], fh, indent=1, sort_keys=True)
    log('Number of files that cannot be read', stats["unread"])
    log('Number of extracted faces', stats["extracted"])
    return stats


@click.command()
@click.option('--config', default=None, type=Path, help='Path to yaml config file with used options of the application.')
def main(**options):
    cfg = config_mod.load_config(options['config'])
    dbase = dataset.Database(cfg.dataset)
    print('input dataset:', dbase)
    print('output directory', cfg.outdir)
    detector = FaceDetector(detector=cfg.detector if cfg.detector else 'pypimtcnn', weights_file=cfg.mtcnn.weights_file)
    print(detector)
    extract_faces(dbase.classes, Path(cfg.outdir).expanduser(), detector, cfg.image, bool(cfg.detect_multiple_faces))


if __name__ == '__main__':
    main()
