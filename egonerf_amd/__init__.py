"""egonerf_amd — MI355X-native (gfx950) implementation of EgoNeRF's volume-rendering hot path.

Public surface mirrors the reference modules for that path:
    egonerf_amd.model       <- models/EgoNeRF.py, models/tensorBase.py, models/envmap.py
    egonerf_amd.coordinates <- models/coordinates.py (YinYangSphericalCoords)
    egonerf_amd.renderer    <- renderer.py (volume_renderer, PSNR evaluation, ray sharding)
    egonerf_amd.sampler     <- sampler.py
Native code: egonerf_amd/csrc/*.hip -> libegonerf_hip.so behind include/egonerf_hip.h (C ABI).
"""
__all__ = ["model", "coordinates", "renderer", "sampler", "synth"]
