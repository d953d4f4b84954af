"""Shared test plumbing: build the product model / the oracle from one synthetic weight set."""
import contextlib
import io

import numpy as np
import torch

from egonerf_amd import synth


def make_coords(cfg, device):
    from egonerf_amd.coordinates import YinYangSphericalCoords
    return YinYangSphericalCoords(device, cfg.aabb, exp_r=True, N_voxel=cfg.n_voxel, r0=cfg.r0, interval_th=True)


def make_model(cfg, weights, device="cuda"):
    """egonerf_amd EgoNeRF with the reference's ctor kwargs (train.py:163-171 resolved values)."""
    from egonerf_amd.model import EgoNeRF
    coords = make_coords(cfg, device)
    assert coords.resolution == cfg.grid
    model = EgoNeRF(torch.from_numpy(cfg.aabb), cfg.grid, device, coords, density_n_comp=list(cfg.density_n_comp),
                    appearance_n_comp=list(cfg.app_n_comp), app_dim=cfg.app_dim, near_far=[cfg.near, cfg.far],
                    shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=cfg.density_shift,
                    distance_scale=cfg.distance_scale, pos_pe=6, view_pe=cfg.view_pe, fea_pe=cfg.fea_pe, featureC=cfg.featureC,
                    step_ratio=0.5, fea2denseAct="softplus", use_envmap=cfg.use_envmap, envmap_res_H=cfg.envmap_res_H,
                    coarse_sigma_grid_update_rule="conv", coarse_sigma_grid_reso=None, interval_th=True)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items() if k != "envmap.emission"})
    if cfg.use_envmap:
        model.envmap.load_envmap(weights["envmap.emission"], device=device)
    if torch.device(device).type == "cuda":
        model.update_coarse_sigma_grid()
    model.eval()
    return model


def make_oracle(cfg, weights, dtype=torch.float32):
    from oracle.egonerf_oracle import OracleScene
    return OracleScene(cfg, weights, dtype=dtype)


def maxerr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
