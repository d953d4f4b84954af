"""Host-side cost of one EgoNeRF.forward call (eval) and what it means for the reference's default chunk of 4096 rays:
ms per call for a tiny batch (GPU time negligible -> host time) and rays/s of a 2^18-ray render at several chunk sizes."""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from egonerf_amd.renderer import volume_renderer

dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
kw = dict(n_coarse=128, n_fine=128, exp_sampling=True, resampling=True, use_coarse_sample=True)
out = {}
with torch.no_grad():
    tiny = torch.from_numpy(synth.make_rays(32, seed=3)).to(dev)
    for _ in range(200):
        model(tiny, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        model(tiny, need_alpha=False, **kw)
    torch.cuda.synchronize()
    out["host_ms_per_forward_call"] = (time.perf_counter() - t0)
    rays = torch.from_numpy(synth.make_rays(1 << 18, seed=4)).to(dev)
    for chunk in (4096, 16384, 65536):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            volume_renderer(rays, model, chunk=chunk, device=dev, keep_alpha=False, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[f"chunk_{chunk}_Mrays_per_s"] = round(rays.shape[0] / dt / 1e6, 2)
print(json.dumps(out))
