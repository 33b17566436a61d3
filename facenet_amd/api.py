"""Inference API of the reference's ``facenet`` package (facenet/__init__.py:16-84): ``FaceNet(config)``,
``.evaluate(images)``, ``.image_to_embedding(arrs)``, ``.embedding_size``.

The reference binds tensor names of a frozen TF graph ('input:0' -> 'embeddings:0', 'phase_train:0' fed False);
there is no TF here, so ``config.path`` names a weights file in Keras variable layout (``.npz`` written by
``InceptionResnetV1.save_weights``; a directory must hold exactly one, like tfutils.py:286-294).  ``config.output``
keeps its meaning: 'embeddings:0' (default when ``config.normalize``) is the L2-normalised output, anything else the
un-normalised bottleneck."""
from __future__ import annotations

from pathlib import Path
from typing import Iterable

import numpy as np

nodes = {   # facenet/__init__.py:16-27 (dtype enums replaced by numpy dtypes)
    "input": {"name": "input", "type": np.uint8},
    "output": {"name": "embeddings", "type": np.float32},
}
config_nodes = {"image_size": {"name": "image_size:0", "type": np.uint8}}   # :29-34


class FaceNet:
    def __init__(self, config):
        from .config import Config
        from .facenet import ImageProcessing, inputs
        from .models.inception_resnet_v1 import InceptionResnetV1, default_config

        if not config.input:
            config.input = nodes["input"]["name"] + ":0"
        if not config.output:
            if config.normalize:
                config.output = nodes["output"]["name"] + ":0"
            else:
                config.output = "InceptionResnetV1/Bottleneck/BatchNorm/Reshape_1:0"
        self._normalized = config.output == nodes["output"]["name"] + ":0"
        image = config.image if config.image else Config({"size": 160, "normalization": 0})
        model_cfg = dict(default_config)
        if config.embedding_size:
            model_cfg["output"] = {"size": int(config.embedding_size)}
        self._model = InceptionResnetV1(inputs(image), ImageProcessing(image), Config(model_cfg),
                                        device=config.device if config.device else "cuda")
        if config.path:
            path = Path(config.path).expanduser()
            if path.is_dir():
                files = list(path.glob("*.npz"))
                if len(files) != 1:
                    raise ValueError("There should not be more than one npz file in the model directory {}.".format(path))
                path = files[0]
            if not path.exists():
                raise ValueError("Model file {} does not exist".format(path))
            self._model.load_weights(path)

    @property
    def embedding_size(self):
        return self._model.embedding_size

    def evaluate(self, images):
        emb = self._model(images, training=False)
        if not self._normalized:   # un-normalised bottleneck (BN output in inference mode)
            n = emb.shape[0]
            plan = self._model._plan(n, False)
            emb = plan.embedding.buf.act.view(n, -1)
        return emb.detach().cpu().numpy()

    def image_to_embedding(self, image_arrays: Iterable[np.ndarray]) -> np.ndarray:
        image_arrays = np.asarray(image_arrays)
        if image_arrays.ndim == 3:
            image_arrays = np.expand_dims(image_arrays, 0)
        return self.evaluate(image_arrays)
