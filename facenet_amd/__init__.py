"""MI355X-native FaceNet hot path.  Mirrors the public surface of the reference's
``facenet`` package (facenet/__init__.py:37-84) -- filled in by facenet_amd.api (imported on first use, so that the
decode workers of facenet_amd.dataset can import facenet_amd._decode without loading torch)."""


def __getattr__(name):
    if name in ("FaceNet", "nodes", "config_nodes"):
        from . import api
        return getattr(api, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
