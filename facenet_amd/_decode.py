"""Host half of ImageLoader (facenet.py:49-50: read_file + decode_image(channels=3)).  No torch import: this is what the
decode worker processes of facenet_amd.dataset load."""
import numpy as np


def decode(path) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def decode_many(paths):
    return [decode(p) for p in paths]


_SHM = {}      # worker-side cache of attached shared-memory blocks


def decode_into(shm_name: str, stride: int, first: int, paths):
    """Decode `paths` into the shared staging block `shm_name`: image k goes to byte offset (first + k) * stride.  Returns the
    (h, w) of every image; an image that does not fit its stride is returned as an array instead (the parent packs it)."""
    from multiprocessing import shared_memory
    shm = _SHM.get(shm_name)
    if shm is None:
        shm = shared_memory.SharedMemory(name=shm_name)
        _SHM[shm_name] = shm
    out = []
    for k, p in enumerate(paths):
        a = decode(p)
        if a.size > stride:
            out.append((a.shape[0], a.shape[1], a))
            continue
        off = (first + k) * stride
        np.frombuffer(shm.buf, dtype=np.uint8, count=a.size, offset=off)[:] = a.reshape(-1)
        out.append((a.shape[0], a.shape[1], None))
    return out
