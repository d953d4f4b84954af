"""Why does bench.py's alt_early_termination step take longer than the default step?  Times EgoNeRF.forward with / without eps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egonerf_amd import synth
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), "cuda")
rays = torch.from_numpy(synth.make_rays(4096, seed=1)).cuda()
kw = dict(n_coarse=512, exp_sampling=True)
def t(n=100):
    with torch.no_grad():
        for _ in range(20): model(rays, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): model(rays, **kw)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for eps in (0.0, 1e-5, 0.0, 1e-5):
    model.early_termination_eps = eps
    print("eps", eps, "ms/step", round(t(), 4), flush=True)
for na in (True, False):
    with torch.no_grad():
        for _ in range(20): model(rays, need_alpha=na, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): model(rays, need_alpha=na, **kw)
        torch.cuda.synchronize(); print("need_alpha", na, round((time.perf_counter() - t0) / 100 * 1e3, 4))
