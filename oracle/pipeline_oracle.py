"""CPU restatement of the input-pipeline pieces on the step's left edge.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in facenet_amd/).  PARITY UNPINNED: the reference ships no
vectors for its loader, and the crop/pad lives in a third-party dependency that is not in /root/reference
(tensorflow~=2.4.0, requirements.txt:1); what follows restates the published algorithm of
tf.image.resize_with_crop_or_pad as the reference calls it (facenet/facenet.py:45-54):

    width_diff  = target_width  - width          offset_crop_width  = max(-width_diff  // 2, 0)
    height_diff = target_height - height         offset_pad_width   = max( width_diff  // 2, 0)
    cropped = crop_to_bounding_box(image, offset_crop_height, offset_crop_width, min(th, h), min(tw, w))
    resized = pad_to_bounding_box(cropped, offset_pad_height, offset_pad_width, th, tw)        (zero padding)
"""
import random

import numpy as np


def resize_with_crop_or_pad(image: np.ndarray, target_height: int, target_width: int) -> np.ndarray:
    h, w = image.shape[:2]
    wd, hd = target_width - w, target_height - h
    ocw, och = max(-wd // 2, 0), max(-hd // 2, 0)
    opw, oph = max(wd // 2, 0), max(hd // 2, 0)
    ch, cw = min(target_height, h), min(target_width, w)
    out = np.zeros((target_height, target_width) + image.shape[2:], image.dtype)
    out[oph:oph + ch, opw:opw + cw] = image[och:och + ch, ocw:ocw + cw]
    return out


def equal_batches(class_files, nrof_classes_per_batch, nrof_examples_per_class, rng: random.Random):
    """One draw of the P x K generator of dataset.py:73-82: `random.sample` of classes, then of each class's files."""
    files, indexes = [], []
    for idx in rng.sample(range(len(class_files)), nrof_classes_per_batch):
        files += rng.sample(class_files[idx], nrof_examples_per_class)
        indexes += [idx] * nrof_examples_per_class
    return files, indexes
