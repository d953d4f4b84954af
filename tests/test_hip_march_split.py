"""ego_march_density with a ray's passes dealt to two waves (few rounds of wave slots, long rays: the 4096 x 512 headline) must return
the bits of the one-wave-per-ray form: weights, background weight, alpha, distances, tile flags (the coordinates of samples behind an
exactly opaque prefix are the one documented difference: the one-wave form never evaluates them and writes zeros)."""
import os

import numpy as np
import pytest
import torch

from egonerf_amd import _lib, synth

pytestmark = pytest.mark.gpu


def march(model, cfg, rays, S, with_alpha, term_eps=0.0):
    lib, st = _lib.load(), _lib.stream_handle()
    model.early_termination_eps = term_eps
    sc = model.scene()
    N = rays.shape[0]
    dev = rays.device
    sched = model._sched(S, dev)
    z, w, bg = torch.zeros(N, S, device=dev), torch.zeros(N, S, device=dev), torch.zeros(N, device=dev)
    crd = torch.zeros(N, S, 4, device=dev)
    alpha = torch.zeros(N, S + 1, device=dev) if with_alpha else None
    act = torch.zeros((N * S + 31) // 32, device=dev, dtype=torch.uint8)
    _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), _lib.ptr(alpha), S + 1 if with_alpha else 0,
                                     w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, act.data_ptr(), st), "march")
    torch.cuda.synchronize()
    return z, w, bg, alpha, act, crd


@pytest.mark.parametrize("S,n_rays,density_shift", [(512, 4096, None), (300, 1001, None), (256, 130, 0.0), (449, 67, 0.0)])
@pytest.mark.parametrize("with_alpha", [False, True])
def test_two_waves_per_ray_return_the_same_bits(S, n_rays, density_shift, with_alpha):
    cfg = synth.SceneConfig(n_voxel=40 ** 3) if density_shift is None else synth.SceneConfig(n_voxel=40 ** 3, density_shift=density_shift)
    model = synth.build_model(cfg, synth.make_weights(cfg, seed=7), "cuda")
    rays = torch.from_numpy(synth.make_rays(n_rays, seed=5)).cuda()
    out = {}
    try:
        for split in ("1", "2"):
            os.environ["EGO_MARCH_SPLIT"] = split
            out[split] = march(model, cfg, rays, S, with_alpha, term_eps=1e-4 if S == 449 else 0.0)
    finally:
        os.environ.pop("EGO_MARCH_SPLIT", None)
    a, b = out["1"], out["2"]
    for k, name in enumerate(("z", "weight", "bg", "alpha", "tile flags")):
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), (name, S, n_rays)
    assert float(a[1].sum()) > 0.0
    shaded = a[1] > 0
    assert torch.equal(a[5][shaded], b[5][shaded])   # coordinates wherever a colour is read
    if density_shift is not None and not with_alpha:   # opaque field: the one-wave form stops behind exact zero transmittance
        assert bool((a[1] == 0).any())


@pytest.mark.parametrize("kw", [dict(n_coarse=512), dict(n_coarse=64, n_fine=64, resampling=True), dict(n_coarse=96)])
@pytest.mark.parametrize("prec", ["f16f6", "f16f8", "f16x3"])
@pytest.mark.parametrize("envmap", [False, True])
def test_folded_compositing_equals_the_two_launch_form(kw, prec, envmap):
    """EGO_RENDER_FOLD=1: ego_render_forward shades and composites in one launch (ego_shade_composite) wherever the scene allows it;
    =0: ego_shade + ego_composite (the default picks by balance: ego_render_forward_folds).  Same products AND - since round 5, when
    ego_composite took over the folded kernel's summation order - the same sums: all five outputs bit for bit, so a batch renders identically
    whichever form its size selects."""
    cfg = synth.SceneConfig(n_voxel=40 ** 3, use_envmap=envmap, envmap_res_H=64) if envmap else synth.SceneConfig(n_voxel=40 ** 3)
    model = synth.build_model(cfg, synth.make_weights(cfg, seed=11), "cuda")
    model.mlp_precision = prec
    rays = torch.from_numpy(synth.make_rays(333, seed=9)).cuda()
    out = {}
    try:
        for fold in (True, False):
            os.environ["EGO_RENDER_FOLD"] = "1" if fold else "0"
            with torch.no_grad():
                out[fold] = model(rays, exp_sampling=True, **kw)
    finally:
        os.environ.pop("EGO_RENDER_FOLD", None)
    for k in range(5):
        a, b = out[True][k], out[False][k]
        assert (a is None) == (b is None), k
        if a is not None:
            assert torch.equal(a, b), (k, float((a - b).abs().max()))
    assert float(out[True][0].max()) > 0.05


def test_default_fold_decision_follows_the_balance_of_whole_rays():
    """ego_render_forward_folds: the folded kernel deals whole rays to its 2048 waves, so it is taken by default only where that costs
    nothing against the tile-granular two-launch form (round 5: it is 3.5 % faster there); the environment overrides either way."""
    from egonerf_amd import _lib
    lib = _lib.load()
    cfg = synth.SceneConfig(n_voxel=40 ** 3)
    model = synth.build_model(cfg, synth.make_weights(cfg, seed=11), "cuda")
    sc = model.scene()
    old = os.environ.pop("EGO_RENDER_FOLD", None)
    try:
        q = lambda N, S: int(lib.ego_render_forward_folds(sc, N, S))
        assert q(4096, 512) == 1 and q(16384, 256) == 1 and q(8192, 256) == 1 and q(1 << 21, 256) == 1
        assert q(4097, 512) == 0 and q(333, 512) == 0 and q(256, 64) == 0 and q(4096, 500) == 0 and q(0, 512) == 0
        os.environ["EGO_RENDER_FOLD"] = "0"
        assert q(4096, 512) == 0
        os.environ["EGO_RENDER_FOLD"] = "1"
        assert q(333, 512) == 1 and q(4096, 500) == 0     # forced, but only where the scene and the sample count qualify
        model.mlp_precision = "f32"
        assert int(lib.ego_render_forward_folds(model.scene(), 4096, 512)) == 0
    finally:
        os.environ.pop("EGO_RENDER_FOLD", None)
        if old is not None:
            os.environ["EGO_RENDER_FOLD"] = old
