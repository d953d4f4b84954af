"""GPU: parity of the inference arithmetics on TRAINED weights.  Every other parity test uses synthetic weights (smooth tables,
uniform-init linears); the `f16f8` arithmetic carries the correction terms of the MLP products in e4m3 (saturating above
448, flushing below 2^-9 of the block scale), so it is also checked on a model whose tables and MLP came out of the optimiser:
a freshly initialised student (0.1 * randn tables, default nn.Linear init, like train.py) fitted to a synthetic teacher scene
with the graphed training iteration, then rendered against the CPU oracle built from the student's state dict."""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from egonerf_amd.model import EgoNeRF
from egonerf_amd.optim import FusedAdam
from egonerf_amd.renderer import volume_renderer
from egonerf_amd.train import GraphedTrainStep
from tests.helpers import make_model, make_oracle, maxerr

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(n_coarse=32, n_fine=32, exp_sampling=True, resampling=True, use_coarse_sample=True)


def _student(cfg):
    return EgoNeRF(torch.from_numpy(cfg.aabb), cfg.grid, DEV, synth.build_coords(cfg, DEV), density_n_comp=list(cfg.density_n_comp),
                   appearance_n_comp=list(cfg.app_n_comp), app_dim=cfg.app_dim, near_far=[cfg.near, cfg.far], shadingMode="MLP_Fea",
                   alphaMask_thres=1e-4, density_shift=cfg.density_shift, distance_scale=cfg.distance_scale, pos_pe=6,
                   view_pe=cfg.view_pe, fea_pe=cfg.fea_pe, featureC=cfg.featureC, step_ratio=0.5, fea2denseAct="softplus",
                   coarse_sigma_grid_update_rule="conv", interval_th=True)


def test_inference_arithmetics_on_trained_weights():
    torch.manual_seed(0)
    cfg = synth.SceneConfig(n_voxel=30 ** 3, density_shift=-4.0)     # a fairly opaque teacher: surfaces, not fog
    teacher = make_model(cfg, synth.make_weights(cfg, seed=77, mlp_gain=4.0), DEV)
    pool, batch, iters = 32768, 2048, 400
    rays_all = torch.from_numpy(synth.make_rays(pool + 512, seed=21)).to(DEV)
    with torch.no_grad():
        rgb_all = volume_renderer(rays_all, teacher, chunk=16384, device=DEV, keep_alpha=False, **KW)[0]
    student = _student(cfg)
    student.train()
    student.update_coarse_sigma_grid()
    opt = FusedAdam(student.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), capturable=True, lr_factor=0.1 ** (1 / iters))
    held = slice(pool, pool + 512)

    def held_out_mse():
        with torch.no_grad():
            out = volume_renderer(rays_all[held], student, chunk=512, device=DEV, keep_alpha=False, **KW)[0]
        return float(((out - rgb_all[held]) ** 2).mean())

    mse0 = held_out_mse()
    g = torch.Generator(device=DEV).manual_seed(1)
    idx = torch.randint(0, pool, (batch,), device=DEV, generator=g)
    step = GraphedTrainStep(student, opt, rays_all[idx], rgb_all[idx], KW, warmup=2)
    for _ in range(iters):
        idx = torch.randint(0, pool, (batch,), device=DEV, generator=g)
        step(rays_all[idx], rgb_all[idx])
    torch.cuda.synchronize()
    student.eval()
    mse1 = held_out_mse()
    assert mse1 < 0.2 * mse0, (mse0, mse1)                            # the fit worked: these are trained weights
    w_max = max(float(p.detach().abs().max()) for p in student.renderModule.parameters())
    # the oracle on the student's state dict (reference layout)
    weights = {k: v.detach().cpu().numpy() for k, v in student.state_dict().items()}
    oracle = make_oracle(cfg, weights)
    rays = rays_all[held][:256]
    ref = oracle.forward(rays.cpu(), **KW)
    errs = {}
    for prec in ("f16f6", "f16f8", "f16x3", "f32"):
        student.mlp_precision = prec
        with torch.no_grad():
            got = student(rays, **KW)
        errs[prec] = (maxerr(got[0], ref[0]), maxerr(got[1], ref[1]))
    print("trained-weight parity (max |d rgb|, max |d depth|):", errs, "largest MLP weight", w_max, "held-out mse", mse0, "->", mse1)
    for prec, (e_rgb, e_d) in errs.items():
        assert e_rgb <= 1e-4 and e_d <= 1e-3 * cfg.far, (prec, e_rgb, e_d)
