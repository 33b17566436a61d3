"""Host half of ImageLoader (facenet.py:49-50: read_file + decode_image(channels=3)).  No torch import: this is what the
decode worker processes of facenet_amd.dataset load."""
import numpy as np


def decode(path) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def decode_many(paths):
    return [decode(p) for p in paths]
