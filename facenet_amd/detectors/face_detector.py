"""Mirror of facenet/detectors/face_detector.py (BoundingBox, image_processing, FaceDetector) with the 'pypimtcnn' detector
served by facenet_amd.detectors.mtcnn instead of the PyPI package.  The Faster-RCNN detector of the reference
(`detectors/frcnnv3`, a frozen TF1 graph whose weights are absent from the reference tree) is out of scope."""
from __future__ import annotations

import math

import numpy as np
from PIL import Image

from . import mtcnn as _mtcnn


def image_processing(image, box, options):
    """face_detector.py:9-26: crop the box plus a relative margin, resize to size * (1 + margin) with PIL's antialias filter."""
    if not isinstance(image, Image.Image):
        raise ValueError('Input must be PIL.Image')
    dw, dh = round(box.width * options.margin / 2), round(box.height * options.margin / 2)
    side = math.ceil(options.size + options.size * options.margin)
    window = (box.left - dw, box.top - dh, box.right + dw, box.bottom + dh)
    return image.crop(window).resize((side, side), getattr(Image, "LANCZOS", None) or Image.ANTIALIAS)   # ANTIALIAS == LANCZOS


class BoundingBox:
    """face_detector.py:29-60: integer box with an exclusive right / bottom edge."""

    def __init__(self, left, top, width, height, confidence=None):
        self.left, self.top = int(np.round(left)), int(np.round(top))
        self.right, self.bottom = int(np.round(left + width)) + 1, int(np.round(top + height)) + 1
        self.width, self.height = self.right - self.left - 1, self.bottom - self.top - 1
        self.confidence = confidence

    def info(self, mode=False):
        fields = [self.left, self.top, self.width, self.height, self.confidence]
        if mode:
            return "left = {}, top = {}, width = {}, height = {}, confidence = {}".format(*fields)
        return str(fields)

    __repr__ = lambda self: self.info(mode=True)
    left_upper = property(lambda self: (self.left, self.top))
    right_lower = property(lambda self: (self.right, self.bottom))
    confidence_as_string = property(lambda self: str(np.round(self.confidence, 3)))


class MTCNN:
    """face_detector.py:63-78."""

    def __init__(self, **kwargs):
        self.__detector = _mtcnn.MTCNN(**kwargs).detect_faces
        self.mode = 'RGB'

    def detector(self, image):
        return [BoundingBox(left=f['box'][0], top=f['box'][1], width=f['box'][2], height=f['box'][3], confidence=f['confidence'])
                for f in self.__detector(image)]


class FaceDetector:
    """face_detector.py:98-123; `detector='pypimtcnn'` is the only one built (keyword arguments go to the MTCNN constructor)."""

    def __init__(self, detector='pypimtcnn', gpu_memory_fraction=1.0, **kwargs):
        if detector == 'frcnnv3':
            raise NotImplementedError("frcnnv3 (frozen Faster-RCNN graph, weights absent from the reference) is out of scope")
        if detector != 'pypimtcnn':
            raise ValueError('Undefined face detector type {}'.format(detector))
        backend = MTCNN(**kwargs)
        self.detector, self.mode, self._find = detector, backend.mode, backend.detector

    def detect(self, image):
        """image: uint8 array [height, width, 3] in `self.mode` channel order -> list of BoundingBox."""
        return self._find(image)

    def __repr__(self):
        return f'class {type(self).__name__}\ndetector type: {self.detector}'
