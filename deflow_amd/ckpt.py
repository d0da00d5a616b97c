"""Reading checkpoints written by the reference's trainer ([REF deflow.py:41-47]: ``torch.load(path)["state_dict"]``).

A real Lightning checkpoint of the reference (``deflow_best.ckpt``) carries more than tensors: ``hyper_parameters["cfg"]`` is
an omegaconf ``DictConfig`` and the callback states reference pytorch_lightning classes.  Neither package is in this image,
and torch >= 2.6 unpickles with ``weights_only=True`` by default, so a plain ``torch.load`` fails on exactly the files this
plugin exists to consume.  ``load_checkpoint`` therefore tries the safe loader first and then falls back to an unpickler
that replaces every class it cannot import by an inert stand-in (attributes kept, no code run), which is enough to reach
``state_dict`` and to read the configuration back as plain Python containers (``plain``).
"""
from __future__ import annotations

import importlib
import pickle
from typing import Any, Dict

import torch


class _Missing:
    """Stand-in for an instance of a class whose module is not installed: keeps whatever state the pickle carried."""

    def __init__(self, *args, **kwargs):
        self._args, self._kwargs = args, kwargs

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):   # (dict, slots) form
            if isinstance(state[0], dict):
                self.__dict__.update(state[0])
            self.__dict__.update(state[1])
        else:
            self._state = state

    # containers pickled through __reduce_ex__ feed items this way
    def append(self, x):
        self.__dict__.setdefault("_items", []).append(x)

    def extend(self, xs):
        self.__dict__.setdefault("_items", []).extend(xs)

    def __setitem__(self, k, v):
        self.__dict__.setdefault("_map", {})[k] = v

    def __call__(self, *a, **k):       # enum-like lookups, functools.partial targets
        return _Missing(*a, **k)


_STUB_CACHE: Dict[tuple, type] = {}


def _stub_class(module: str, name: str) -> type:
    key = (module, name)
    if key not in _STUB_CACHE:
        _STUB_CACHE[key] = type(name, (_Missing,), {"__module__": module, "_df_stub": True})
    return _STUB_CACHE[key]


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            mod = importlib.import_module(module)
            obj = mod
            for part in name.split("."):
                obj = getattr(obj, part)
            return obj
        except Exception:
            return _stub_class(module, name)


class _TolerantPickle:
    """the ``pickle_module`` interface torch.load expects"""
    __name__ = "deflow_amd_tolerant_pickle"
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())


def load_checkpoint(path: str) -> Dict[str, Any]:
    """-> the checkpoint dict (map_location="cpu").  Tensors-only files take the safe path; anything else is read with
    unknown classes stubbed.  Only open checkpoints you trust: the fallback is a full unpickle."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)


def plain(obj: Any) -> Any:
    """Configuration objects -> plain dict / list / scalar.  Understands omegaconf's pickled layout through the stand-ins
    (``DictConfig`` / ``ListConfig``: ``_content`` holding nodes; value nodes: ``_val``) and ordinary containers."""
    if getattr(type(obj), "_df_stub", False):
        d = obj.__dict__
        if "_content" in d:
            return plain(d["_content"])
        if "_val" in d:
            return plain(d["_val"])
        if "_map" in d:
            return plain(d["_map"])
        if "_items" in d:
            return plain(d["_items"])
        if "_value_" in d:            # enum member
            return plain(d["_value_"])
        return {k: plain(v) for k, v in d.items() if not k.startswith("_")}
    if isinstance(obj, dict):
        return {str(k): plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [plain(v) for v in obj]
    if hasattr(obj, "items") and callable(obj.items) and not isinstance(obj, torch.Tensor):   # a live DictConfig
        try:
            return {str(k): plain(v) for k, v in obj.items()}
        except Exception:
            pass
    return obj


def flatten(cfg: Any, prefix: str = "") -> Dict[str, Any]:
    """{"model": {"target": {"num_iters": 4}}} -> {"model.target.num_iters": 4}; flat keys pass through.  Lists stay
    values (``voxel_size``, ``point_cloud_range``)."""
    out: Dict[str, Any] = {}
    if not isinstance(cfg, dict):
        return out
    for k, v in cfg.items():
        key = f"{prefix}{k}"
        if isinstance(v, dict):
            out.update(flatten(v, key + "."))
        else:
            out[key] = v
    return out
