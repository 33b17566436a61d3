"""Helpers on the hot path, same names and behaviour as facenet/facenet.py:
``inputs`` (:35-36), ``ImageProcessing`` (:57-86), ``evaluate_embeddings`` (:184-201),
``LearningRateScheduler`` (:381-400)."""
from __future__ import annotations

import numpy as np
import torch


def inputs(config):
    """facenet.py:35-36 returns ``tf.keras.Input([size, size, 3])``; here the symbolic shape itself."""
    return (config.size, config.size, 3)


class ImageProcessing:
    """Input normalisation layer (facenet.py:57-86).  It carries the configuration; the arithmetic runs
    in fn_image_normalize as the first launch of every plan (fused there with the channel padding)."""

    def __init__(self, config):
        self.input_node_name = "input"
        self.config = config
        self.image_size = (config.size, config.size)
        self.eps = 1e-3
        if config.normalization not in (0, 1):
            raise ValueError("Invalid image normalization algorithm")   # facenet.py:82


def evaluate_embeddings(model, dset):
    """facenet.py:184-201: run ``model(images)`` over (images, labels) batches and concatenate."""
    embeddings_, labels_ = [], []
    for images, labels in dset:
        emb = model(images)
        embeddings_.append(emb.detach().cpu().numpy() if torch.is_tensor(emb) else np.asarray(emb))
        labels_.append(np.asarray(labels))
    return np.concatenate(embeddings_), np.concatenate(labels_)


class LearningRateScheduler:
    """facenet.py:381-400 (0-based epoch; constant when ``config.value`` is set)."""

    def __init__(self, config):
        self.config = config
        self.default_value = self.config.value if self.config.value else None

    def __call__(self, epoch):
        if self.default_value is not None:
            return self.default_value
        learning_rate = self.config.schedule[-1][1]
        for (epoch_, learning_rate) in self.config.schedule:
            if epoch < epoch_:
                break
        return learning_rate
