"""Reading checkpoints written by the reference (`EgoNeRF.save`, models/EgoNeRF.py:158-172).

A reference `.th` file is `torch.save({'kwargs', 'state_dict', 'global_step', 'alphaMask_{yin,yang}.{shape,mask}',
'envmap.emission', 'envmap_res_H'})`, and `kwargs` pickles live objects of `models.coordinates.YinYangSphericalCoords`
and `models.envmap.EnvironmentMap` (models/tensorBase.py:241-268).  Unpickling therefore needs those module paths;
`reference_module_shims()` provides them (only for the duration of the load, and only if the real reference is not
importable) by mapping them onto this package's classes, whose `__setstate__` accept the reference's attribute sets.
"""
from __future__ import annotations

import contextlib
import sys
import types

import torch


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
