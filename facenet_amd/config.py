"""Attribute-access config mirroring facenet/config.py:54-107 (missing attributes are an empty, falsy
Config).  Only the keys the hot path reads are interpreted."""
from __future__ import annotations

from pathlib import Path


_INDENT = "   "     # the reference prints nested sections three spaces deeper (facenet/config.py:72)


def _wrap(value):
    return Config(value) if isinstance(value, dict) else value


def _unwrap(value):
    return value.as_dict if isinstance(value, Config) else value


def _lines(fields, depth):
    """YAML-like listing of a field table: 'key: value' lines, a nested section as 'key: ' plus its own lines indented."""
    for key, value in fields.items():
        if isinstance(value, Config):
            yield f"{_INDENT * depth}{key}: "
            yield from _lines(value._fields, depth + 1)
        else:
            yield f"{_INDENT * depth}{key}: {value}"


class Config:
    """Settings tree with attribute access.  Contract of facenet/config.py:54-107: nested dicts become nested Configs, a
    missing attribute reads as an empty (falsy) Config so that `if cfg.a.b.c:` never raises, attributes may be assigned,
    `as_dict` / `items()` / `exists()` expose the fields.  The fields live in one private table instead of the instance
    dictionary."""

    __slots__ = ("_fields",)

    def __init__(self, dct=None):
        object.__setattr__(self, "_fields", {k: _wrap(v) for k, v in (dct or {}).items()})

    def __getattr__(self, name):            # only reached for names that are not slots / methods
        if name.startswith("_"):            # copy / pickle protocol probes (and a missing field table) are real errors
            raise AttributeError(name)
        found = self._fields.get(name)
        return Config() if found is None and name not in self._fields else found

    def __setattr__(self, name, value):
        self._fields[name] = _wrap(value)

    def __reduce__(self):                   # copy / pickle rebuild through the constructor
        return (Config, (self.as_dict,))

    def __bool__(self):
        return len(self._fields) > 0

    def __repr__(self):
        return "".join(line + "\n" for line in _lines(self._fields, 0))

    @property
    def as_dict(self):
        return {k: _unwrap(v) for k, v in self._fields.items()}

    def items(self):
        return self._fields.items()

    def exists(self, name):
        return name in self._fields


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
