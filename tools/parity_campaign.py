"""Randomised parity campaign: the HIP render vs the CPU oracle over random scene seeds, grids, ray / sample counts,
resampling modes and envmap settings (eval mode; tolerance of north_star: 1e-4 RGB).  Prints the worst errors."""
import sys, os, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from egonerf_amd.synth import build_model
from oracle.egonerf_oracle import OracleScene

dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
worst = dict(rgb=0.0, depth=0.0, alpha=0.0)
torch.set_num_threads(16)
for case in range(n_cases):
    nv = int(rng.choice([20, 24, 30, 40])) ** 3
    env = bool(rng.integers(0, 2))
    scene = dict(rng.choice([dict(near=0.01, far=15.0, r0=0.03, density_shift=-8.0), dict(near=0.1, far=300.0, r0=0.05, density_shift=-10.0),
                             dict(near=0.01, far=50.0, r0=0.05, density_shift=-8.0)]))
    cfg = synth.SceneConfig(n_voxel=nv, use_envmap=env, envmap_res_H=int(rng.choice([8, 16, 33])), **scene)
    w = synth.make_weights(cfg, seed=int(rng.integers(1, 10 ** 6)))
    model, oracle = build_model(cfg, w, dev), OracleScene(cfg, w)
    N = int(rng.choice([1, 7, 64, 130, 257]))
    rays = torch.from_numpy(synth.make_rays(N, seed=int(rng.integers(1, 10 ** 6))))
    resampling = bool(rng.integers(0, 2))
    kw = dict(n_coarse=int(rng.choice([5, 24, 33, 64, 100])), n_fine=int(rng.choice([2, 16, 37, 64])) if resampling else 0,
              resampling=resampling, use_coarse_sample=bool(rng.integers(0, 2)) if resampling else True)
    if resampling and kw["n_coarse"] < 4:
        kw["n_coarse"] = 8
    with torch.no_grad():
        got = model(rays.to(dev), exp_sampling=True, **kw)
        ref = oracle.forward(rays, **kw)
    e_rgb = float((got[0].cpu() - ref[0]).abs().max())
    e_dep = float((got[1].cpu() - ref[1]).abs().max()) / max(float(ref[1].abs().max()), 1.0)
    e_alpha = float((got[4].cpu() - ref[4]).abs().max()) if not resampling else 0.0
    worst = dict(rgb=max(worst["rgb"], e_rgb), depth=max(worst["depth"], e_dep), alpha=max(worst["alpha"], e_alpha))
    flag = "" if e_rgb <= 1e-4 else "   <-- ABOVE TOLERANCE"
    print(f"case {case:2d}: grid {cfg.grid} env {int(env)} near/far {cfg.near}/{cfg.far} N {N:3d} {kw}  rgb {e_rgb:.2e} depth(rel) {e_dep:.2e} alpha {e_alpha:.2e}{flag}")
print("worst:", worst)
sys.exit(0 if worst["rgb"] <= 1e-4 else 1)
