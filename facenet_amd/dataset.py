"""Input pipeline on the left edge of the step (reference: facenet/dataset.py, facenet/facenet.py:45-54).

Directory-per-class listing (`ImageClass`, `Database`), the shuffled batch iterator (`tf_dataset_api`) and the P x K
identity sampler (`pipeline_with_equal_batches`) keep the reference's names, attributes and errors.  What changes is
where the pixels are handled: a thread pool decodes files to ragged HWC uint8 arrays, a batch is packed back to back
into pinned host memory, copied on a side stream and centre-cropped / zero-padded on the GPU
(`fn_crop_or_pad_u8` = tf.image.resize_with_crop_or_pad), so what reaches the step is the uint8 NHWC batch its first
kernel (`fn_image_normalize`) reads.  Batches are prefetched `prefetch` deep; the consumer's stream waits on an event.

Not carried over: the `h5file` validity filter (h5py is not in this image: a config that
sets it raises) and tqdm/loguru progress output.
"""
from __future__ import annotations

import random
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from . import _lib
from ._decode import decode as _decode, decode_into as _decode_into, decode_many as _decode_many


class ImageLoader:
    """facenet.py:45-54: read_file + decode_image(channels=3) + resize_with_crop_or_pad(size, size).  `decode` is the host
    half (PIL), the crop/pad runs on the GPU; `loader(path)` returns the uint8 [size, size, 3] device tensor."""

    def __init__(self, config=None):
        self.height = config.size
        self.width = config.size
        if self.height != self.width or not self.height:
            raise ValueError("ImageLoader: config.size must be a positive integer")

    decode = staticmethod(_decode)

    def __call__(self, path):
        return crop_or_pad_batch([self.decode(path)], self.height)[0]


def crop_or_pad_batch(arrays, size: int, device="cuda", stream=None, staging=None) -> torch.Tensor:
    """Ragged list of HWC uint8 arrays -> uint8 [N, size, size, 3] on the device (one H2D copy + one launch)."""
    lib = _lib.load()
    n = len(arrays)
    if n == 0:
        return torch.empty(0, size, size, 3, dtype=torch.uint8, device=device)
    hw = np.empty((n, 2), np.int32)
    off = np.empty(n, np.int64)
    total = 0
    for i, a in enumerate(arrays):
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
            raise ValueError(f"image {i}: expected HWC uint8 with 3 channels, got {a.dtype} {a.shape}")
        hw[i] = a.shape[:2]
        off[i] = total
        total += a.size
    meta = 8 * n + 8 * n                                   # offsets (i64) then hw (2 x i32) in front of the pixels
    if staging is None or staging.numel() < meta + total:
        staging = torch.empty(meta + total, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else \
            torch.empty(meta + total, dtype=torch.uint8)
    host = staging.numpy()
    host[:8 * n] = off.view(np.uint8)
    host[8 * n:meta] = hw.reshape(-1).view(np.uint8)
    for i, a in enumerate(arrays):
        host[meta + off[i]:meta + off[i] + a.size] = a.reshape(-1)
    st = stream if stream is not None else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        dev = staging[:meta + total].to(device, non_blocking=True)
        out = torch.empty(n, size, size, 3, dtype=torch.uint8, device=device)
        _lib.check(lib.fn_crop_or_pad_u8(dev.data_ptr() + meta, dev.data_ptr(), dev.data_ptr() + 8 * n, out.data_ptr(), n, size,
                                         st.cuda_stream))
    out._staging = staging          # kept alive (and reusable) until the batch is dropped
    return out


class ImageClass:
    """Stores the paths to images for a given class (dataset.py:104-142)."""

    def __init__(self, config):
        if not config.path:
            raise ValueError("Path to download dataset does not specified.")
        self.path = Path(config.path).expanduser()
        self.name = self.path.stem
        if not self.path.exists():
            raise ValueError(f"Directory {self.path} does not exist")
        files = [f for f in self.path.glob("*") if f.is_file()]
        if config.h5file:
            raise NotImplementedError("h5file validity filter: h5py is not available in this build")
        if config.max_nrof_images and len(files) > config.max_nrof_images:
            files = list(np.random.choice(files, size=config.max_nrof_images, replace=False))
        self.files = sorted(str(f) for f in files)

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.name}/{self.nrof_images})"

    @property
    def nrof_images(self):
        return len(self.files)

    @property
    def nrof_pairs(self):
        return self.nrof_images * (self.nrof_images - 1) // 2


class Database:
    """Directory-per-class data set (dataset.py:145-231)."""

    def __init__(self, config):
        if not config.path:
            raise ValueError("Path to download dataset does not specified.")
        self.path = Path(config.path).expanduser()
        if not self.path.exists():
            raise ValueError(f"Directory {self.path} does not exist")
        self.h5file = Path(config.h5file).expanduser() if config.h5file else config.h5file
        dirs = [p for p in self.path.glob("*") if p.is_dir()]
        if config.nrof_classes and len(dirs) > config.nrof_classes:
            dirs = list(np.random.choice(dirs, size=config.nrof_classes, replace=False))
        dirs.sort()
        self.classes = []
        for path in dirs:
            config.path = path                      # the reference mutates the config the same way (dataset.py:170)
            images = ImageClass(config)
            if images.nrof_images > 0:
                self.classes.append(images)

    def __repr__(self):
        return (f"{self.__class__.__name__}\n" + f"{self.path}\n" + f"h5 file {self.h5file}\n" +
                f"Number of classes {self.nrof_classes} \n" + f"Number of images {self.nrof_images}\n" +
                f"Minimal number of images in class {self.min_nrof_images}\n" +
                f"Maximal number of images in class {self.max_nrof_images}\n")

    @property
    def files(self):
        return [f for cls in self.classes for f in cls.files]

    @property
    def labels(self):
        return np.array([idx for idx, cls in enumerate(self.classes) for _ in range(cls.nrof_images)])

    @property
    def min_nrof_images(self):
        return min(cls.nrof_images for cls in self.classes)

    @property
    def max_nrof_images(self):
        return max(cls.nrof_images for cls in self.classes)

    @property
    def nrof_classes(self):
        return len(self.classes)

    @property
    def nrof_images(self):
        return sum(cls.nrof_images for cls in self.classes)

    @property
    def nrof_images_per_class(self):
        return [cls.nrof_images for cls in self.classes]

    def tf_dataset_api(self, loader, batch_size, buffer_size=None, repeat=False, **kw):
        return tf_dataset_api(self.files, self.labels, loader, batch_size, buffer_size=buffer_size, repeat=repeat, **kw)


class BatchPipeline:
    """Iterable of (uint8 [B,size,size,3] device tensor, int64 [B] device tensor).

    `plan()` yields (files, labels) per batch; decode runs in `workers` threads -- or, with `processes=True`, in that
    many spawned worker processes (PIL decode holds the GIL for part of its work; processes scale with the cores) --
    `prefetch` batches are in flight, the H2D copy and the crop/pad launch go to a side stream and the consumer's current
    stream waits on the batch's event.  Worker processes never touch the GPU (they import facenet_amd._decode only)."""

    CHUNK = 10      # files per worker task in process mode

    def __init__(self, plan, loader: ImageLoader, cardinality=None, workers=8, prefetch=2, device="cuda", processes=False,
                 max_image_bytes: int = 300 * 300 * 3):
        self._plan, self.loader, self._card = plan, loader, cardinality
        self.workers, self.prefetch, self.device, self.processes = workers, max(1, prefetch), device, processes
        self.max_image_bytes = (int(max_image_bytes) + 15) // 16 * 16
        self._pool = None
        self._slots = []          # process mode: shared-memory staging blocks (pinned), one per batch in flight

    # ---- process mode: workers decode straight into pinned shared memory --------------------------------------------------
    def _new_slot(self, nbytes: int):
        from multiprocessing import shared_memory
        shm = shared_memory.SharedMemory(create=True, size=nbytes)
        t = torch.frombuffer(shm.buf, dtype=torch.uint8)
        rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), nbytes, 0)      # pinned: the H2D copy can be asynchronous
        return {"shm": shm, "tensor": t, "pinned": int(rc) == 0, "bytes": nbytes}

    def _free_slots(self):
        for s in self._slots:
            try:
                if s["pinned"]:
                    torch.cuda.cudart().cudaHostUnregister(s["tensor"].data_ptr())
                s["tensor"] = None
                s["shm"].close()
                s["shm"].unlink()
            except Exception:
                pass
        self._slots = []

    def _executor(self):
        if self._pool is None:
            if self.processes:
                import multiprocessing
                from concurrent.futures import ProcessPoolExecutor
                self._pool = ProcessPoolExecutor(self.workers, mp_context=multiprocessing.get_context("spawn"))
            else:
                self._pool = ThreadPoolExecutor(self.workers)
        return self._pool

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=self.processes, cancel_futures=True)
            self._pool = None
        self._free_slots()

    def __del__(self):
        self.close()

    def cardinality(self):
        return self._card          # None = infinite (tf.data.INFINITE_CARDINALITY)

    def __len__(self):
        if self._card is None:
            raise TypeError("infinite pipeline")
        return self._card

    def _iter_shared(self):
        """Process mode with the stock decoder: each batch owns a shared-memory slot with a fixed stride per image; the workers
        write pixels into it and return only (h, w), the parent issues ONE H2D copy of the slot and the crop/pad launch.
        Nothing is pickled or repacked on the way (the parent thread was the bottleneck at ~11 000 images/s before)."""
        lib = _lib.load()
        pool = self._executor()
        side = torch.cuda.Stream(device=self.device)
        stride, size = self.max_image_bytes, self.loader.height
        pending, free = deque(), list(range(len(self._slots)))
        plan = iter(self._plan())

        def submit():
            nonlocal stride
            try:
                files, labels = next(plan)
            except StopIteration:
                return False
            stride = self.max_image_bytes
            need = stride * len(files)
            k = next((i for i in free if self._slots[i]["bytes"] >= need), None)
            if k is None:
                self._slots.append(self._new_slot(max(need, stride * 128)))
                k = len(self._slots) - 1
            else:
                free.remove(k)
            name = self._slots[k]["shm"].name
            futs = [pool.submit(_decode_into, name, stride, i, files[i:i + self.CHUNK]) for i in range(0, len(files), self.CHUNK)]
            pending.append((futs, labels, k, len(files), stride))
            return True

        try:
            for _ in range(self.prefetch):
                if not submit():
                    break
            while pending:
                futs, labels, k, n, stride = pending.popleft()
                res = [r for f in futs for r in f.result()]
                submit()
                slot = self._slots[k]
                if any(arr is not None for _, _, arr in res):
                    # an image larger than the stride came back as an array: this batch takes the packing path, and the
                    # stride grows so that later batches fit (slots are re-created on demand)
                    arrays = [arr if arr is not None else
                              np.frombuffer(slot["shm"].buf, np.uint8, count=h * w * 3, offset=i * stride).reshape(h, w, 3).copy()
                              for i, (h, w, arr) in enumerate(res)]
                    images = crop_or_pad_batch(arrays, size, self.device, side)
                    del images._staging
                    with torch.cuda.stream(side):
                        lab = torch.as_tensor(np.asarray(labels, np.int64)).to(self.device)
                    self.max_image_bytes = stride = (max(a.size for a in arrays) + 15) // 16 * 16
                else:
                    hw = torch.tensor([[h, w] for h, w, _ in res], dtype=torch.int32)
                    off = torch.arange(n, dtype=torch.int64) * stride
                    with torch.cuda.stream(side):
                        dev = slot["tensor"][:n * stride].to(self.device, non_blocking=slot["pinned"])
                        d_hw, d_off = hw.to(self.device), off.to(self.device)
                        images = torch.empty(n, size, size, 3, dtype=torch.uint8, device=self.device)
                        _lib.check(lib.fn_crop_or_pad_u8(dev.data_ptr(), d_off.data_ptr(), d_hw.data_ptr(), images.data_ptr(), n, size,
                                                         side.cuda_stream))
                        lab = torch.as_tensor(np.asarray(labels, np.int64)).to(self.device)
                done = torch.cuda.Event()
                done.record(side)
                torch.cuda.current_stream().wait_event(done)
                for t in (images, lab):
                    t.record_stream(torch.cuda.current_stream())
                done.synchronize()                              # the slot is reusable once its copy has run
                free.append(k)
                yield images, lab
        finally:
            for futs, *_ in pending:
                for f in futs:
                    f.cancel()

    def __iter__(self):
        if self.processes and type(self.loader).decode is _decode:
            yield from self._iter_shared()
            return
        pool = self._executor()
        decode_many = type(self.loader).decode is _decode and self.processes
        side = torch.cuda.Stream(device=self.device)
        pending, free = deque(), []
        plan = iter(self._plan())

        def submit():
            try:
                files, labels = next(plan)
            except StopIteration:
                return False
            if decode_many:
                futs = [pool.submit(_decode_many, files[i:i + self.CHUNK]) for i in range(0, len(files), self.CHUNK)]
            else:
                futs = [pool.submit(self.loader.decode, f) for f in files]
            pending.append((futs, labels))
            return True

        try:
            for _ in range(self.prefetch):
                if not submit():
                    break
            while pending:
                futs, labels = pending.popleft()
                arrays = [a for f in futs for a in f.result()] if decode_many else [f.result() for f in futs]
                submit()
                need = sum(a.size for a in arrays) + 16 * len(arrays)
                k = next((i for i, s in enumerate(free) if s.numel() >= need), None)
                staging = free.pop(k) if k is not None else None
                images = crop_or_pad_batch(arrays, self.loader.height, self.device, side, staging)
                with torch.cuda.stream(side):
                    lab = torch.as_tensor(np.asarray(labels, np.int64)).to(self.device, non_blocking=True)
                done = torch.cuda.Event()
                done.record(side)
                torch.cuda.current_stream().wait_event(done)
                images.record_stream(torch.cuda.current_stream())
                lab.record_stream(torch.cuda.current_stream())
                done.synchronize()                  # the pinned staging buffer is free again once the copy has run
                free.append(images._staging)
                del images._staging
                yield images, lab
        finally:
            for futs, _ in pending:
                for f in futs:
                    f.cancel()


def tf_dataset_api(files, labels, loader, batch_size, buffer_size=None, repeat=False, **kw):
    """dataset.py:15-43: zip(files, labels) -> optional shuffle -> optional repeat -> batch -> prefetch.  With a
    `buffer_size` the reference shuffles once globally and then through a buffer_size*batch_size window, reshuffled
    every iteration; here every epoch is a fresh full permutation (the window's limit case).  `repeat` comes before
    `batch` as in the reference, so a repeating stream has full batches that span epochs and only a finite one ends short."""
    files, labels = list(files), list(np.asarray(labels).tolist())
    if len(files) != len(labels):
        raise ValueError("files and labels differ in length")
    n = len(files)

    def plan():
        bf, bl = [], []
        while True:
            order = np.random.permutation(n) if buffer_size is not None else np.arange(n)
            for i in order:
                bf.append(files[i])
                bl.append(labels[i])
                if len(bf) == batch_size:
                    yield bf, bl
                    bf, bl = [], []
            if not repeat:                       # repeat() sits before batch(): only a finite stream has a short batch
                if bf:
                    yield bf, bl
                return

    card = None if repeat else (n + batch_size - 1) // batch_size
    return BatchPipeline(plan, loader, card, **kw)


def pipeline_with_equal_batches(loader, classes, config, **kw):
    """dataset.py:46-101: endless P x K batches, 20 random classes x 5 random files each (the reference overwrites the
    two config values the same way, :61-62); labels are class indexes.  Raises ValueError (from random.sample) when there
    are fewer classes / files than asked for, as the reference's generator does."""
    config.nrof_classes_per_batch = 20
    config.nrof_examples_per_class = 5
    for idx, _class in enumerate(classes):
        _class.index = idx

    def plan():
        while True:
            _files, _indexes = [], []
            for cls in random.sample(classes, config.nrof_classes_per_batch):
                _files += random.sample(cls.files, config.nrof_examples_per_class)
                _indexes += [cls.index] * config.nrof_examples_per_class
            yield _files, _indexes

    return BatchPipeline(plan, loader, None, **kw)
