"""End-to-end training demo on synthetic data: the loop body of the reference's train.py (:245-330) — SimpleSampler batches,
render, MSE (+ optional TV / L1 / ortho / entropy terms), FusedAdam with the per-step lr decay, coarse-table refresh — fits a
freshly initialised model to rays rendered from a synthetic "ground-truth" scene.  Prints PSNR over the iterations and a
held-out PSNR:   python tools/train_demo.py [--iters 400] [--n-voxel 1000000] [--batch 4096] [--reg]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from egonerf_amd.losses import TVLoss, ray_entropy_loss
from egonerf_amd.model import EgoNeRF
from egonerf_amd.optim import FusedAdam
from egonerf_amd.renderer import volume_renderer
from egonerf_amd.sampler import SimpleSampler

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=400)
ap.add_argument("--n-voxel", type=float, default=100 ** 3)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--pool", type=int, default=262144)
ap.add_argument("--reg", action="store_true", help="add the Ricoh configs' regularisers (TV 0.1 / 0.01, L1 8e-5, ortho 0, entropy 1e-3)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = synth.SceneConfig(n_voxel=a.n_voxel)
teacher = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
kw = dict(n_coarse=64, n_fine=64, exp_sampling=True, resampling=True, use_coarse_sample=True, interval_th=True)
rays_all = torch.from_numpy(synth.make_rays(a.pool + 8192, seed=5)).to(dev)
with torch.no_grad():
    rgb_all = volume_renderer(rays_all, teacher, chunk=65536, device=dev, keep_alpha=False, **kw)[0]
train_rays, train_rgb, test_rays, test_rgb = rays_all[: a.pool], rgb_all[: a.pool], rays_all[a.pool:], rgb_all[a.pool:]

torch.manual_seed(0)
np.random.seed(20221028)  # train.py:413
student = EgoNeRF(torch.from_numpy(cfg.aabb), cfg.grid, dev, synth.build_coords(cfg, dev), density_n_comp=list(cfg.density_n_comp),
                  appearance_n_comp=list(cfg.app_n_comp), app_dim=cfg.app_dim, near_far=[cfg.near, cfg.far], shadingMode="MLP_Fea",
                  alphaMask_thres=1e-4, density_shift=cfg.density_shift, distance_scale=cfg.distance_scale, pos_pe=6,
                  view_pe=cfg.view_pe, fea_pe=cfg.fea_pe, featureC=cfg.featureC, step_ratio=0.5, fea2denseAct="softplus",
                  coarse_sigma_grid_update_rule="conv", interval_th=True)   # fresh 0.1 * randn tables, default nn.Linear init
student.train()
opt = FusedAdam(student.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))       # lr_init / lr_basis of configs/EgoNeRF/common.txt
lr_factor = 0.1 ** (1 / a.iters)                                                   # lr_decay_target_ratio over the run (train.py:176-182)
sampler, tv = SimpleSampler(a.pool, a.batch), TVLoss()
tv_d, tv_a, ent_w = 0.1, 0.01, 1e-3


def psnr(mse):
    return -10.0 * np.log(mse) / np.log(10.0)


def held_out():
    with torch.no_grad():
        out = volume_renderer(test_rays, student, chunk=8192, device=dev, keep_alpha=False, **kw)[0]
    return psnr(float(((out - test_rgb) ** 2).mean()))


log = [dict(iter=0, test_psnr=held_out())]
torch.cuda.synchronize()
t0 = time.perf_counter()
recent = []
for it in range(a.iters):
    idx = sampler.nextids().to(dev)
    rgb_map, _, _, _, alpha = volume_renderer(train_rays[idx], student, chunk=a.batch, device=dev, is_train=True, **kw)
    loss = torch.mean((rgb_map - train_rgb[idx]) ** 2)
    total = loss
    if a.reg:
        tv_d *= lr_factor; tv_a *= lr_factor; ent_w *= lr_factor
        total = total + 8e-5 * student.density_L1() + tv_d * student.TV_loss_density(tv) + tv_a * student.TV_loss_app(tv) \
            + ent_w * ray_entropy_loss(alpha)
    opt.zero_grad()
    total.backward()
    opt.step()
    for g in opt.param_groups:
        g["lr"] *= lr_factor
    student.update_coarse_sigma_grid()   # train.py:356-357
    recent.append(loss.item())
    if (it + 1) % max(a.iters // 8, 1) == 0:
        log.append(dict(iter=it + 1, train_psnr=psnr(float(np.mean(recent))), test_psnr=held_out()))
        recent = []
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(config=f"train demo: grid {cfg.grid}, {a.batch} rays x (64+64), {a.iters} iterations, regularisers {a.reg}",
                      s_total=dt, ms_per_iter_incl_eval=dt / a.iters * 1e3, log=log)))
