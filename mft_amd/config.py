"""Attribute-bag configuration with the semantics the tracker relies on
(``MFT/config.py:8-52``): any missing attribute reads as an empty, falsy
``Config`` (so ``C.timers_enabled`` / ``C.cache_delta_infinity`` default to
"off"), and a config file is a Python file exposing ``get_config()``."""
from __future__ import annotations

import importlib.util
from pathlib import Path


class Config:
    def __getattr__(self, name):
        # only reached for attributes that were never set
        if name.startswith("__"):
            raise AttributeError(name)
        return Config()

    def __bool__(self):
        return False

    def merge(self, other, update_dicts=False):
        for key, value in other.__dict__.items():
            mine = self.__dict__.get(key)
            if update_dicts and isinstance(mine, dict) and isinstance(value, dict):
                mine.update(value)
            else:
                setattr(self, key, value)

    def __repr__(self):
        return repr(self.__dict__)

    def __eq__(self, other):
        return isinstance(other, self.__class__) and self.__dict__ == other.__dict__


def load_config(path):
    path = Path(path)
    assert path.exists(), f"config {path} does not exist!"
    spec = importlib.util.spec_from_file_location("tracker_config", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.get_config()


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__.update(kwargs)
