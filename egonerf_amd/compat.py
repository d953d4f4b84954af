"""Reading checkpoints written by the reference (`EgoNeRF.save`, models/EgoNeRF.py:158-172).

A reference `.th` file is `torch.save({'kwargs', 'state_dict', 'global_step', 'alphaMask_{yin,yang}.{shape,mask}',
'envmap.emission', 'envmap_res_H'})`, and `kwargs` pickles live objects of `models.coordinates.YinYangSphericalCoords`
and `models.envmap.EnvironmentMap` (models/tensorBase.py:241-268).  Unpickling therefore needs those module paths;
`reference_module_shims()` provides them (only for the duration of the load, and only if the real reference is not
importable) by mapping them onto this package's classes, whose `__setstate__` accept the reference's attribute sets.

Writing goes the other way round: inside `reference_pickle_paths()` (which `EgoNeRF.save` enters) this package's
`YinYangSphericalCoords` / `EnvironmentMap` objects pickle as `models.coordinates.YinYangSphericalCoords` /
`models.envmap.EnvironmentMap` with exactly the reference's attribute sets, so the reference's own
`torch.load` + `eval(model_name)(**ckpt['kwargs'])` (train.py:52-56) opens a file written here.
"""
from __future__ import annotations

import contextlib
import copyreg
import sys
import types

import torch

_PICKLE_AS_REFERENCE = 0   # depth of active reference_pickle_paths() contexts
_REF_CLASSES = {}          # (module, name) -> class object pickled by reference while a context is active


def pickling_as_reference() -> bool:
    return _PICKLE_AS_REFERENCE > 0


def reduce_as_reference(module: str, name: str, state: dict):
    """__reduce_ex__ value that re-creates an attribute-only object of `module.name`: the stdlib's protocol-0/1 object
    reconstructor (object.__new__(cls), then BUILD = __dict__.update(state) for a class without __setstate__)."""
    return copyreg._reconstructor, (_REF_CLASSES[(module, name)], object, None), state


@contextlib.contextmanager
def reference_pickle_paths():
    """While active, `models.coordinates.YinYangSphericalCoords` and `models.envmap.EnvironmentMap` resolve (pickle looks classes
    up by module path and checks identity) either to the real reference classes, when the reference is imported in this
    process, or to empty stand-in classes of the same module path; this package's objects then reduce to those."""
    global _PICKLE_AS_REFERENCE
    targets = (("models.coordinates", "YinYangSphericalCoords"), ("models.envmap", "EnvironmentMap"))
    saved = {n: sys.modules.get(n) for n in ("models", "models.coordinates", "models.envmap")}
    installed = []
    entered = False
    try:
        for mod, name in targets:
            m = sys.modules.get(mod)
            if m is None or not hasattr(m, name):
                if "models" not in sys.modules:
                    pkg = types.ModuleType("models")
                    pkg.__path__ = []
                    sys.modules["models"] = pkg
                    installed.append("models")
                m = types.ModuleType(mod)
                setattr(m, name, type(name, (), {"__module__": mod}))
                sys.modules[mod] = m
                installed.append(mod)
            _REF_CLASSES[(mod, name)] = getattr(m, name)
        _PICKLE_AS_REFERENCE += 1
        entered = True
        yield
    finally:
        if entered:  # a failure in the set-up above must not leave the counter negative (later saves would then pickle this package's paths)
            _PICKLE_AS_REFERENCE -= 1
        for n in installed:
            if saved[n] is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = saved[n]


@contextlib.contextmanager
def reference_module_shims():
    from . import coordinates, model
    names = ("models", "models.coordinates", "models.envmap")
    saved = {n: sys.modules.get(n) for n in names}
    try:
        if saved["models.coordinates"] is None:
            pkg = types.ModuleType("models")
            pkg.__path__ = []  # mark as package
            mc = types.ModuleType("models.coordinates")
            mc.YinYangSphericalCoords = coordinates.YinYangSphericalCoords
            me = types.ModuleType("models.envmap")
            me.EnvironmentMap = model.EnvironmentMap
            pkg.coordinates, pkg.envmap = mc, me
            sys.modules.update({"models": pkg, "models.coordinates": mc, "models.envmap": me})
        yield
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def load_reference_checkpoint(path: str, device="cuda"):
    """-> (model, global_step).  Mirrors train.py:52-56: `eval(model_name)(**ckpt['kwargs'])` then `model.load(ckpt)`."""
    from .coordinates import YinYangSphericalCoords
    from .model import EgoNeRF
    with reference_module_shims():
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    kwargs = dict(ckpt["kwargs"])
    coords = kwargs.pop("coordinates")
    if not isinstance(coords, YinYangSphericalCoords):
        raise TypeError(f"checkpoint coordinates are {type(coords).__name__}; only the yin-yang grid is supported")
    coords.device = device
    aabb, grid = kwargs.pop("aabb"), kwargs.pop("gridSize")
    envmap = kwargs.pop("envmap", None)
    kwargs.update(device=device, envmap=envmap, interval_th=coords.interval_th)
    model = EgoNeRF(aabb, grid, coordinates=coords, **kwargs)
    step = model.load(ckpt)
    return model, step
