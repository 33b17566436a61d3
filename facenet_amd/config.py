"""Attribute-access config mirroring facenet/config.py:54-107 (missing attributes are an empty, falsy
Config).  Only the keys the hot path reads are interpreted."""
from __future__ import annotations

from pathlib import Path


class Config:
    def __init__(self, dct=None):
        if dct is None:
            dct = dict()
        for key, item in dct.items():
            if isinstance(item, dict):
                setattr(self, key, Config(item))
            else:
                setattr(self, key, item)

    def __repr__(self):
        def get_str(obj, ident=""):
            s = ""
            for key, item in obj.items():
                if isinstance(item, Config):
                    s += f"{ident}{key}: \n{get_str(item, ident=ident + '   ')}"
                else:
                    s += f"{ident}{key}: {str(item)}\n"
            return s
        return get_str(self)

    def __getattr__(self, name):
        return self.__dict__.get(name, Config())

    def __bool__(self):
        return bool(self.__dict__)

    @property
    def as_dict(self):
        def as_dict(obj):
            s = {}
            for key, item in obj.items():
                s[key] = as_dict(item) if isinstance(item, Config) else item
            return s
        return as_dict(self)

    def items(self):
        return self.__dict__.items()

    def exists(self, name):
        return name in self.__dict__.keys()


class LoadConfigError(Exception):
    pass


DEFAULTS = {   # apps/configs/config.yaml:4-15 and train_softmax.yaml:37-47 (the keys the hot path reads)
    "seed": 0,
    "batch_size": 100,
    "image": {"size": 160, "margin": 0, "normalization": 0},
    "train": {"epoch": {"nrof_epochs": None, "size": 1000},
              "learning_rate": {"value": None, "schedule": [[100, 0.05], [200, 0.005], [300, 0.0005]]}},
    "loss": {"alpha": 0.2},
}


def _merge(a: dict, b: dict) -> dict:
    out = dict(a)
    for k, v in b.items():
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def load_config(path=None, overrides: dict = None) -> Config:
    """YAML merged over the defaults (config.py:114-142 merge order collapsed to defaults <- file <- overrides)."""
    cfg = dict(DEFAULTS)
    if path is not None:
        import yaml
        path = Path(path).expanduser()
        if not path.is_file():
            raise LoadConfigError(f"config file {path} does not exist")
        with open(path) as f:
            cfg = _merge(cfg, yaml.safe_load(f) or {})
    if overrides:
        cfg = _merge(cfg, overrides)
    c = Config(cfg)
    if not c.train.epoch.nrof_epochs:           # config.py:181-182
        c.train.epoch.nrof_epochs = c.train.learning_rate.schedule[-1][0]
    return c
