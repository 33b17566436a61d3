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


def train_softmax(cfg, nrof_classes: int, batches=None, embedding_size: int = 512, device: str = "cuda", use_graph: bool = True,
                  world_size: int = 1, process_group=None, log=print):
    net = Network(embedding_size=embedding_size, image_size=cfg.image.size, normalization=cfg.image.normalization,
                  nrof_classes=nrof_classes, device=device, seed=cfg.seed)
    scheduler = LearningRateScheduler(cfg.train.learning_rate)
    trainer = Trainer(net, batch=cfg.batch_size, loss="softmax", lr=scheduler(0), world_size=world_size, process_group=process_group)
    if batches is None:
        batches = synthetic_batches(cfg.batch_size, nrof_classes, cfg.image.size, cfg.seed)
    batches = iter(batches)
    if use_graph:
        x, y = next(batches)
        trainer.set_images(x, y)
        trainer.capture()
    for epoch in range(cfg.train.epoch.nrof_epochs):
        trainer.set_learning_rate(scheduler(epoch))              # Keras LearningRateScheduler callback: 0-based epoch
        t0 = time.perf_counter()
        for _ in range(cfg.train.epoch.size):
            x, y = next(batches)
            trainer.set_images(x, y)
            trainer.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        log(f"epoch {epoch + 1}/{cfg.train.epoch.nrof_epochs}  xent {trainer.loss_value():.4f}  lr {scheduler(epoch)}  "
            f"{cfg.batch_size * cfg.train.epoch.size * world_size / dt:.1f} img/s")
        if cfg.model.path:                                        # ModelCheckpoint(save_weights_only=True) each epoch (:74-78)
            path = Path(cfg.model.path).expanduser()
            path.mkdir(parents=True, exist_ok=True)
            np.savez(path / f"{path.stem}.npz", **{k: v.numpy() for k, v in net.export_keras_params().items()})
    return net, trainer


@click.command()
@click.option("--config", default=None, type=Path, help="Path to yaml config file with used options of the application.")
@click.option("--nrof-classes", default=10575, type=int, help="Number of identities (synthetic data when no dataset is wired in).")
def main(**options):
    cfg = config_mod.load_config(options["config"])
    if cfg.dataset.path:                                          # apps/train_softmax.py:28-47
        from facenet_amd import dataset
        loader = dataset.ImageLoader(config=cfg.image)
        train_dbase = dataset.Database(cfg.dataset)
        batches = train_dbase.tf_dataset_api(loader=loader, batch_size=cfg.batch_size, repeat=True, buffer_size=10, processes=True)
        train_softmax(cfg, train_dbase.nrof_classes, batches)
    else:
        train_softmax(cfg, options["nrof_classes"])


if __name__ == "__main__":
    main()
