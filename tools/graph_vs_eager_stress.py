"""Distribution of the eager-vs-graph differences that tests/test_hip_train_graph.py bounds (float atomics + Adam): repeats the
test's six iterations `reps` times and prints, per repetition, the worst loss difference, per-tensor outlier counts and the final
render difference.   python tools/graph_vs_eager_stress.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egonerf_amd import synth
from egonerf_amd.optim import FusedAdam
from egonerf_amd.train import GraphedTrainStep
from tests.helpers import make_model
DEV = "cuda"
KW = dict(n_coarse=16, n_fine=16, exp_sampling=True, resampling=True, use_coarse_sample=True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, factor = 192, 0.9
batches = [(torch.from_numpy(synth.make_rays(N, seed=40 + i)).to(DEV),
            torch.from_numpy(synth.hash_uniform(70 + i, 0, N * 3).reshape(N, 3).astype(np.float32)).to(DEV)) for i in range(6)]
jit = torch.from_numpy(synth.hash_uniform(9, 0, N * 16).reshape(N, 16).astype(np.float32)).to(DEV)
def setup():
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    m = make_model(cfg, synth.make_weights(cfg, seed=3), DEV); m.train(); m.update_coarse_sigma_grid(); return m
worst = dict(loss=0.0, mean=0.0, n_off=0, frac_off=0.0, maxerr=0.0, render=0.0)
for rep in range(reps):
    # some unrelated GPU work first (allocator / clock state like inside the full suite)
    junk = [torch.randn(1 << (18 + (rep + k) % 5), device=DEV).sin_().sum() for k in range(4)]
    m_ref = setup()
    o_ref = FusedAdam(m_ref.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    ref_losses = []
    for rays, gt in batches:
        rgb, *_ = m_ref(rays, is_train=True, jitter=jit, u=jit, **KW)
        loss = torch.mean((rgb - gt) ** 2)
        o_ref.zero_grad(set_to_none=True); loss.backward(); o_ref.step()
        for grp in o_ref.param_groups: grp["lr"] *= factor
        m_ref.update_coarse_sigma_grid(); ref_losses.append(float(loss.detach()))
    m_g = setup()
    o_g = FusedAdam(m_g.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), capturable=True, lr_factor=factor)
    step = GraphedTrainStep(m_g, o_g, batches[0][0], batches[0][1], KW, warmup=1, noise_fn=lambda n, mm, dev: jit)
    got = [float(step(r, g)) for r, g in batches[1:]]
    dl = max(abs(a - b) / max(abs(a), 1e-3) for a, b in zip(ref_losses[1:], got))
    pr, pg = dict(m_ref.named_parameters()), dict(m_g.named_parameters())
    line = []
    for k in pr:
        a, b = pr[k].detach(), pg[k].detach()
        scale = max(float(a.abs().max()), 1e-3); err = (a - b).abs()
        n_off = int((err > 2e-4 * scale).sum())
        worst["mean"] = max(worst["mean"], float(err.mean()) / scale); worst["n_off"] = max(worst["n_off"], n_off)
        worst["frac_off"] = max(worst["frac_off"], n_off / err.numel()); worst["maxerr"] = max(worst["maxerr"], float(err.max()))
        if n_off: line.append((k, n_off, err.numel(), round(float(err.max()), 5)))
    m_ref.eval(); m_g.eval()
    with torch.no_grad():
        ra = m_ref(batches[0][0], n_coarse=32, exp_sampling=True)[0]; rb = m_g(batches[0][0], n_coarse=32, exp_sampling=True)[0]
    rd = float((ra - rb).abs().max())
    worst["loss"] = max(worst["loss"], dl); worst["render"] = max(worst["render"], rd)
    print(f"rep {rep}: rel loss diff {dl:.2e} render diff {rd:.2e} outliers {line}")
print("worst:", worst)
