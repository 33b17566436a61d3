"""MI355X-native FaceNet hot path.  Mirrors the public surface of the reference's
``facenet`` package (facenet/__init__.py:37-84) -- filled in by facenet_amd.api."""
from .api import FaceNet, nodes, config_nodes  # noqa: F401
