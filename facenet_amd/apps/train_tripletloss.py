# coding:utf-8
"""Triplet-loss training entry point (the north-star path; the reference only mentions it at README.md:18 -- build-defined,
SURVEY.md A13): ``python -m facenet_amd.apps.train_tripletloss --config x.yaml``.

Per step: embed a P x K pool with the inference path -> [PK,PK] squared distances -> online selection (alpha) ->
train on the selected (a,p,n) rows: forward(training=True) -> l2_normalize -> triplet loss -> backward -> Keras Adam.
Pools come from ``pools`` (an iterable of (uint8 images [P*K,160,160,3])) with labels repeat(arange(P), K) or, by
default, from a seeded synthetic generator."""
from __future__ import annotations

import time
from pathlib import Path

import click
import numpy as np
import torch

from facenet_amd import config as config_mod
from facenet_amd.engine import Network
from facenet_amd.facenet import LearningRateScheduler
from facenet_amd.train import GraphRunner, Trainer, TripletMiner
from facenet_amd.schedule import make_events


def train_tripletloss(cfg, people_per_batch: int = 45, images_per_person: int = 4, nrof_triplets: int = 30, embedding_size: int = 128,
                      pools=None, device: str = "cuda", use_graph: bool = True, world_size: int = 1, process_group=None, log=print):
    alpha = cfg.loss.alpha if cfg.loss.alpha else 0.2
    net = Network(embedding_size=embedding_size, image_size=cfg.image.size, normalization=cfg.image.normalization, device=device, seed=cfg.seed)
    scheduler = LearningRateScheduler(cfg.train.learning_rate)
    trainer = Trainer(net, batch=3 * nrof_triplets, loss="triplet", alpha=alpha, lr=scheduler(0), world_size=world_size,
                      process_group=process_group)
    n = people_per_batch * images_per_person
    miner = TripletMiner(net, n, np.repeat(np.arange(people_per_batch), images_per_person), nrof_triplets, alpha=alpha, seed=cfg.seed)
    miner.build(trainer.plan.images)
    if pools is None:
        g = torch.Generator().manual_seed(cfg.seed)
        pools = (torch.randint(0, 256, (n, cfg.image.size, cfg.image.size, 3), dtype=torch.uint8, generator=g) for _ in iter(int, 1))
    pools = iter(pools)
    mine_graph = None
    if use_graph:
        miner.plan.images.copy_(next(pools))
        miner.run()
        ev = make_events(miner.sched)
        mine_graph = GraphRunner(net.device).capture(lambda: miner.run(ev))
        trainer.capture()                                         # side-effect free (state restored after its warm-up step)
    for epoch in range(cfg.train.epoch.nrof_epochs):
        trainer.set_learning_rate(scheduler(epoch))
        t0 = time.perf_counter()
        for _ in range(cfg.train.epoch.size):
            miner.plan.images.copy_(next(pools))
            mine_graph.replay() if mine_graph is not None else miner.run()
            trainer.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        log(f"epoch {epoch + 1}/{cfg.train.epoch.nrof_epochs}  triplet loss {trainer.loss_value():.4f}  "
            f"{3 * nrof_triplets * cfg.train.epoch.size * world_size / dt:.1f} img/s")
    return net, trainer


@click.command()
@click.option("--config", default=None, type=Path, help="Path to yaml config file with used options of the application.")
def main(**options):
    cfg = config_mod.load_config(options["config"])
    if cfg.dataset.path:      # P x K batches from disk: the reference's equal-batches sampler (dataset.py:46-101), 20 classes x 5 images
        from facenet_amd import dataset
        loader = dataset.ImageLoader(config=cfg.image)
        dbase = dataset.Database(cfg.dataset)
        pipe = dataset.pipeline_with_equal_batches(loader, dbase.classes, cfg, processes=True)
        train_tripletloss(cfg, people_per_batch=cfg.nrof_classes_per_batch, images_per_person=cfg.nrof_examples_per_class,
                          pools=(images for images, _ in pipe))
    else:
        train_tripletloss(cfg)


if __name__ == "__main__":
    main()
