"""How much slower are the any-shape compatibility kernels (csrc/ego_generic.hip) than the tuned path?  ("about an order
of magnitude" deserves a number.)  Same grid, same rays: the shipped shape against shapes that take the compatibility kernels for the head
(shadingMode 'MLP', featureC 64, view_pe = fea_pe = 6) - inference at 4096 x 512 and one training step at 8192 x (128 + 128)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
dev = "cuda"
SHAPES = {"shipped (MFMA path)": {}, "shadingMode MLP (generic head)": dict(shadingMode="MLP"), "featureC 64, small tables (generic)": dict(density_n_comp=(8, 8, 8), app_n_comp=(24, 24, 24), featureC=64),
          "ctor defaults 6 / 6 encodings (generic)": dict(view_pe=6, fea_pe=6, app_dim=12)}
def ms(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, kw in SHAPES.items():
    cfg = synth.SceneConfig(**kw)
    model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
    rays = torch.from_numpy(synth.make_rays(4096, seed=1)).to(dev)
    with torch.no_grad():
        t_inf = ms(lambda: model(rays, n_coarse=512, exp_sampling=True), 10)
    model.train()
    rays8 = torch.from_numpy(synth.make_rays(8192, seed=1)).to(dev); gt = torch.rand(8192, 3, device=dev)
    def step():
        model.zero_grad(set_to_none=True)
        rgb, *_ = model(rays8, is_train=True, n_coarse=128, n_fine=128, exp_sampling=True, resampling=True, use_coarse_sample=True)
        torch.mean((rgb - gt) ** 2).backward()
    t_tr = ms(step, 3)
    print(f"{name:42s} tuned={model.is_tuned_shape!s:5s} inference 4096 x 512: {t_inf:8.3f} ms   training fwd+bwd 8192 x (128+128): {t_tr:8.2f} ms", flush=True)
