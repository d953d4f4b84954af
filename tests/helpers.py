"""Shared test plumbing: build the product model / the oracle from one synthetic weight set."""
import contextlib
import io

import numpy as np
import torch

from egonerf_amd import synth


def make_coords(cfg, device):
    return synth.build_coords(cfg, device)


def make_model(cfg, weights, device="cuda"):
    """egonerf_amd EgoNeRF with the reference's ctor kwargs (train.py:163-171 resolved values)."""
    return synth.build_model(cfg, weights, device)


def make_oracle(cfg, weights, dtype=torch.float32):
    from oracle.egonerf_oracle import OracleScene
    return OracleScene(cfg, weights, dtype=dtype)


def maxerr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def campaign_cases(seed: int, n_cases: int):
    """The randomised scene / ray / sampling-mode stream of the parity campaign (tools/parity_campaign.py and
    tests/test_hip_parity_campaign.py draw the SAME cases from a seed): yields (case, cfg, weights, rays [N,6] CPU, kw)."""
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        nv = int(rng.choice([20, 24, 30, 40])) ** 3
        env = bool(rng.integers(0, 2))
        scene = dict(rng.choice([dict(near=0.01, far=15.0, r0=0.03, density_shift=-8.0), dict(near=0.1, far=300.0, r0=0.05, density_shift=-10.0),
                                 dict(near=0.01, far=50.0, r0=0.05, density_shift=-8.0),
                                 dict(near=0.01, far=15.0, r0=0.03, density_shift=0.0)]))   # opaque: most tiles take the exact zero-weight skip
        cfg = synth.SceneConfig(n_voxel=nv, use_envmap=env, envmap_res_H=int(rng.choice([8, 16, 33])), **scene)
        w = synth.make_weights(cfg, seed=int(rng.integers(1, 10 ** 6)))
        N = int(rng.choice([1, 7, 64, 130, 257]))
        rays = torch.from_numpy(synth.make_rays(N, seed=int(rng.integers(1, 10 ** 6))))
        resampling = bool(rng.integers(0, 2))
        kw = dict(n_coarse=int(rng.choice([5, 24, 33, 64, 100])), n_fine=int(rng.choice([2, 16, 37, 64])) if resampling else 0,
                  resampling=resampling, use_coarse_sample=bool(rng.integers(0, 2)) if resampling else True)
        if resampling and kw["n_coarse"] < 4:
            kw["n_coarse"] = 8
        yield case, cfg, w, rays, kw
