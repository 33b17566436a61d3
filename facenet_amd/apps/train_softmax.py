# coding:utf-8
"""Softmax-classifier training entry point, same shape as the reference's apps/train_softmax.py:21-112:
``python -m facenet_amd.apps.train_softmax --config x.yaml``.

model = InceptionResnetV1; network = Sequential([model, Dense(nrof_classes)]) (:55-66); loss =
SparseCategoricalCrossentropy(from_logits=True) (:91); Adam(epsilon=0.1) (:92); LR set per epoch by
LearningRateScheduler (:80-83); ``train.epoch.size`` steps per epoch (:95-104).

Batches come from ``batches`` (an iterable of (uint8 images [N,160,160,3], int labels [N])); ``main`` builds them from
``cfg.dataset.path`` with facenet_amd.dataset (Database + tf_dataset_api, :28-47 of the reference app) or, when no data
set is configured, from a seeded synthetic generator."""
from __future__ import annotations

import time
from pathlib import Path

import click
import numpy as np
import torch

from facenet_amd import config as config_mod
from facenet_amd.engine import Network
from facenet_amd.facenet import LearningRateScheduler
from facenet_amd.train import Trainer


def synthetic_batches(batch_size, nrof_classes, size, seed):
    g = torch.Generator().manual_seed(seed)
    while True:
        yield (torch.randint(0, 256, (batch_size, size, size, 3), dtype=torch.uint8, generator=g),
               torch.randint(0, nrof_classes, (batch_size,), generator=g))


def init_distributed():
    """One process per GPU under ``python -m torch.distributed.run``: (rank, world, process group, device).  Without the
    launcher's environment: a single replica.  ``MirroredStrategy()`` of apps/train_softmax_tf2_gpus.py:49 becomes RCCL
    (backend "nccl") over xGMI; FACENET_DIST_BACKEND=gloo rehearses the same path on one device."""
    import os
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world == 1:
        return 0, 1, None, "cuda"
    import torch.distributed as dist
    backend = os.environ.get("FACENET_DIST_BACKEND", "nccl")
    dev_index = local if backend == "nccl" else local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    return rank, world, dist.group.WORLD, f"cuda:{dev_index}"


def train_softmax(cfg, nrof_classes: int, batches=None, embedding_size: int = 512, device: str = "cuda", use_graph: bool = True,
                  world_size: int = 1, process_group=None, rank: int = 0, log=print):
    """``cfg.batch_size`` is the GLOBAL batch: MirroredStrategy splits it across the replicas
    (apps/train_softmax_tf2_gpus.py:49-108), so every rank trains ``cfg.batch_size // world_size`` images per step with
    per-replica BatchNorm, gradients are averaged over the global batch, and ``batches`` must yield this rank's shard."""
    if cfg.batch_size % world_size:
        raise ValueError(f"batch_size {cfg.batch_size} is not divisible by the {world_size} replicas")
    local_batch = cfg.batch_size // world_size
    net = Network(embedding_size=embedding_size, image_size=cfg.image.size, normalization=cfg.image.normalization,
                  nrof_classes=nrof_classes, device=device, seed=cfg.seed)
    scheduler = LearningRateScheduler(cfg.train.learning_rate)
    trainer = Trainer(net, batch=local_batch, loss="softmax", lr=scheduler(0), world_size=world_size, process_group=process_group)
    first_epoch = 0
    if cfg.model.checkpoint:                                      # network.load_weights(checkpoint) before fit (:68-71)
        ckpt = Path(cfg.model.checkpoint).expanduser()
        ckpt = ckpt / f"{ckpt.stem}.npz" if ckpt.is_dir() else ckpt
        log(f"Restore checkpoint {ckpt}")
        first_epoch = trainer.load_checkpoint(ckpt)
    if batches is None:
        batches = synthetic_batches(local_batch, nrof_classes, cfg.image.size, cfg.seed + rank)    # a different shard per rank
    batches = iter(batches)
    if use_graph:
        x, y = next(batches)
        trainer.set_images(x, y)
        trainer.capture()                                         # side-effect free: no uncounted step, Adam's t untouched
        batches = _chain_first((x, y), batches)                   # the batch used for the capture is trained on as batch 1
    for epoch in range(first_epoch, cfg.train.epoch.nrof_epochs):
        trainer.set_learning_rate(scheduler(epoch))              # Keras LearningRateScheduler callback: 0-based epoch
        t0 = time.perf_counter()
        for _ in range(cfg.train.epoch.size):
            x, y = next(batches)
            trainer.set_images(x, y)
            trainer.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            log(f"epoch {epoch + 1}/{cfg.train.epoch.nrof_epochs}  xent {trainer.loss_value():.4f}  lr {scheduler(epoch)}  "
                f"{cfg.batch_size * cfg.train.epoch.size / dt:.1f} img/s")
        if cfg.model.path:                                        # ModelCheckpoint(save_weights_only=True) each epoch (:74-78)
            path = Path(cfg.model.path).expanduser()
            if rank == 0:
                path.mkdir(parents=True, exist_ok=True)
            # Keras variable names and order + Adam slots, iteration count and epoch; moving statistics averaged over replicas
            trainer.save_checkpoint(path / f"{path.stem}.npz", epoch=epoch + 1)
    return net, trainer


def _chain_first(first, rest):
    yield first
    yield from rest


@click.command()
@click.option("--config", default=None, type=Path, help="Path to yaml config file with used options of the application.")
@click.option("--nrof-classes", default=10575, type=int, help="Number of identities (synthetic data when no dataset is wired in).")
def main(**options):
    cfg = config_mod.load_config(options["config"])
    rank, world, pg, device = init_distributed()                  # python -m torch.distributed.run --nproc-per-node N -m facenet_amd.apps.train_softmax
    kw = dict(device=device, world_size=world, process_group=pg, rank=rank)
    if cfg.dataset.path:                                          # apps/train_softmax.py:28-47
        from facenet_amd import dataset
        loader = dataset.ImageLoader(config=cfg.image)
        train_dbase = dataset.Database(cfg.dataset)
        np.random.seed(cfg.seed + rank)                           # every replica walks its own permutation of the data set
        batches = train_dbase.tf_dataset_api(loader=loader, batch_size=cfg.batch_size // world, repeat=True, buffer_size=10, processes=True)
        train_softmax(cfg, train_dbase.nrof_classes, batches, **kw)
    else:
        train_softmax(cfg, options["nrof_classes"], **kw)


if __name__ == "__main__":
    main()
