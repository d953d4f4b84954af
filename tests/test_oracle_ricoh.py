"""Pins the oracle on BASELINE config 3's scene (Ricoh360: near_far [0.1, 300], r0 0.05, density_shift -10, envmap
3 x 3840 x 1920, full [150,172,516] grid) against vectors captured from the real reference (oracle/capture_golden.py
capture_ricoh / capture_envmap_full -> tests/golden/ricoh.npz, envmap_full.npz).  CPU only."""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from oracle.egonerf_oracle import OracleScene, erp_rays_reference

T = torch.from_numpy


@pytest.fixture(scope="module")
def ricoh(golden):
    fx = golden("ricoh")
    cfg = synth.SceneConfig(**synth.RICOH)
    assert cfg.grid == fx["grid"].tolist()
    return fx, OracleScene(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))


def close(a, b, tol):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    assert a.shape == b.shape and err <= tol, f"max err {err} > {tol}"


def test_scene_scalars(ricoh):
    fx, sc = ricoh
    assert float(sc.far_r) == float(fx["far_r"])
    assert np.array_equal((sc.cfg.near + sc.sample_schedule(128)).numpy(), fx["sched128"])


def test_erp_ray_generator_restatement(ricoh):
    """dataLoader/ray_utils.py:24-40 + :85-113 restated in the oracle: bit-exact vs the reference's rays."""
    fx, _ = ricoh
    H, W = int(fx["H"]), int(fx["W"])
    for k in range(2):
        got = erp_rays_reference(H, W, T(fx["poses"][k]), normalize=True)[T(fx["gen_idx"])]
        assert np.array_equal(got.numpy(), fx[f"gen_rays/{k}"])
        got = erp_rays_reference(H, W, T(fx["poses"][k]), normalize=False)[T(fx["gen_idx"])]
        assert np.array_equal(got.numpy(), fx[f"gen_rays_raw/{k}"])
        assert np.array_equal(fx[f"gen_rays/{k}"][np.searchsorted(fx["gen_idx"], fx["idx"])], fx[f"rays/{k}"])


@pytest.mark.parametrize("k", [0, 1])
def test_config3_render(ricoh, k):
    fx, sc = ricoh
    rays = T(fx[f"rays/{k}"])
    with torch.no_grad():
        o = sc.forward(rays, n_coarse=128, n_fine=128, resampling=True, use_coarse_sample=True)
    close(o[0], fx[f"rs128/{k}/rgb"], 2e-6)
    close(o[1], fx[f"rs128/{k}/depth"], 3e-4)          # z up to 300: 1e-6 relative
    close(o[2], fx[f"rs128/{k}/bg"], 2e-6), close(o[3], fx[f"rs128/{k}/env"], 1e-6)
    close(o[4].sum(-1), fx[f"rs128/{k}/acc_alpha_sum"], 1e-4)
    if k == 0:
        with torch.no_grad():
            o = sc.forward(rays, n_coarse=512)
        close(o[0], fx[f"nr512/{k}/rgb"], 2e-6), close(o[1], fx[f"nr512/{k}/depth"], 3e-4)


@pytest.mark.parametrize("h", [1000, 1920])
def test_envmap_at_shipped_sizes(golden, h):
    fx = golden("envmap_full")
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=h)
    w = synth.make_weights(cfg, seed=3)
    w["envmap.emission"] = synth.white_envmap(int(fx["seed"]), h)
    sc = OracleScene(cfg, w)
    close(sc.envmap_radiance(T(fx["dirs"])), fx[f"radiance/{h}"], 1e-6)
