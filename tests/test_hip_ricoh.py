"""BASELINE config 3 on the GPU: Ricoh360 scene (near_far [0.1, 300], r0 0.05, density_shift -10, envmap 3 x 3840 x 1920, full
[150,172,516] grid), equirectangular 1024 x 2048 renders, against vectors captured from the real reference
(tests/golden/ricoh.npz: its own ERP ray generator + EgoNeRF.forward through volume_renderer; envmap_full.npz).

Tolerances (north_star): RGB 1e-4 absolute, depth 1e-3 * z_max (z_max = 300 here)."""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from egonerf_amd.renderer import erp_rays, shard_bounds, volume_renderer
from tests.helpers import make_model, make_oracle, maxerr

pytestmark = pytest.mark.gpu
DEV = "cuda"
RGB_TOL, DEPTH_TOL = 1e-4, 1e-3 * 300.0
KW = dict(n_coarse=128, n_fine=128, exp_sampling=True, resampling=True, use_coarse_sample=True, device=DEV)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def ricoh(golden):
    fx = golden("ricoh")
    cfg = synth.SceneConfig(**synth.RICOH)
    w = synth.make_weights(cfg, seed=int(fx["seed_weights"]))
    model = make_model(cfg, w, DEV)
    assert model.envmap.emission.shape == (3, 3840, 1920)
    model.oracle = make_oracle(cfg, w)
    return fx, cfg, model


BORDER_EPS = 2e-6  # radians: a few float32 ulps of pi; inside it acos/atan2 rounding (ocml here, sleef in the CPU reference) picks the grid


def assert_matches_reference(model, rays, got, want_rgb, want_depth, kw):
    """got = (rgb, depth, ...) of the HIP render of `rays`; want_* = the reference's.  Every ray must meet the tolerances,
    except rays with a sample on a yin/yang region border (coordinates.py:478-481's inclusive comparisons on float32
    acos/atan2 results): there the grid choice is libm-dependent and the two grids hold independent tables.  Those rays are
    few, each must really have a sample within BORDER_EPS of a border, and (single-pass renders) the HIP result must equal
    the oracle evaluated with the HIP's grid choice."""
    e_rgb = (got[0].cpu() - torch.from_numpy(want_rgb)).abs().amax(1)
    e_d = (got[1].cpu() - torch.from_numpy(want_depth)).abs()
    bad = ((e_rgb > RGB_TOL) | (e_d > DEPTH_TOL)).nonzero().flatten()
    # north_star's PSNR clause, stated directly (renderer.py:156-157): gt = the reference's image + integer-hash noise at ~30 dB;
    # PSNR(HIP, gt) and PSNR(reference, gt) agree to 1e-3 dB over the rays both evaluate on the same grid (the border rays below
    # are a property of the reference's float32 comparisons and are checked against the oracle with the grid choice forced)
    ok = torch.ones(e_rgb.shape[0], dtype=torch.bool)
    ok[bad] = False
    d_psnr, _p_hip, p_ref = synth.delta_psnr(got[0].cpu()[ok].numpy(), want_rgb[ok.numpy()])
    assert 28.0 < p_ref < 36.0 and abs(d_psnr) <= 1e-3, (d_psnr, p_ref)
    if bad.numel() == 0:
        return 0
    assert bad.numel() <= max(2, rays.shape[0] // 200), f"{bad.numel()} rays off: not a border effect"
    orc = model.oracle
    with torch.no_grad():
        _, inter = orc.forward(rays[bad].cpu(), keep=True, **{k: v for k, v in kw.items() if k not in ("exp_sampling", "device")})
        pts = [inter["xyz_coarse"]] + ([inter["xyz_fine"]] if "xyz_fine" in inter else [])
        margin = torch.stack([orc.yin_margin(p).abs().amin(1) for p in pts]).amin(0)
        assert bool((margin < BORDER_EPS).all()), f"rays {bad.tolist()} are off without a border sample (margins {margin.tolist()})"
        if not kw.get("resampling"):
            r = rays[bad]
            xyz, _, _ = model.sample_ray_exp(r[:, :3], r[:, 3:6], is_train=False, N_samples=kw["n_coarse"])
            flags = model.coordinates.from_cartesian(xyz)[..., 6].cpu()
            ref_flags = inter["c7n"][..., 6]
            diff = flags != ref_flags
            assert 0 < int(diff.sum()) <= 2 * bad.numel() and bool((orc.yin_margin(inter["xyz_coarse"])[diff].abs() < BORDER_EPS).all())
            forced = orc.forward(r.cpu(), grid_choice=flags, **{k: v for k, v in kw.items() if k not in ("exp_sampling", "device")})
            assert maxerr(got[0][bad], forced[0]) <= RGB_TOL and maxerr(got[1][bad], forced[1]) <= DEPTH_TOL
    return int(bad.numel())


@pytest.fixture(params=["f16x3", "f16f8", "f16f6", "f32"])
def precision(request, ricoh):
    ricoh[2].mlp_precision = request.param
    yield request.param
    ricoh[2].mlp_precision = "f16f6"


def test_erp_rays_vs_reference_generator(ricoh):
    """ego_erp_rays against get_ray_directions_360 + (normalisation) + get_rays of the reference, two poses, 6602 rays incl.
    two full rows and columns, the poles and the phi = +-pi seam."""
    fx, _, _ = ricoh
    H, W = int(fx["H"]), int(fx["W"])
    gi = T(fx["gen_idx"])
    for k in range(2):
        full = erp_rays(H, W, fx["poses"][k], DEV)
        assert full.shape == (H * W, 6)
        assert maxerr(full[gi], fx[f"gen_rays/{k}"]) <= 1e-6
        assert maxerr(erp_rays(H, W, fx["poses"][k], DEV, normalize=False)[gi], fx[f"gen_rays_raw/{k}"]) <= 1e-6
        win = erp_rays(H, W, fx["poses"][k], DEV, row0=300, n_rows=2)  # a shard of rows is the same rays
        assert torch.equal(win, full[300 * W:302 * W])


@pytest.mark.parametrize("k", [0, 1])
def test_config3_subset_vs_reference(ricoh, precision, k):
    """The reference's rays in, 128+128 resampling and 512-sample variants: rgb / depth / bg / env vs the reference."""
    fx, _, model = ricoh
    rays = T(fx[f"rays/{k}"])
    with torch.no_grad():
        out = volume_renderer(rays, model, chunk=4096, **KW)
    n_border = assert_matches_reference(model, rays, out, fx[f"rs128/{k}/rgb"], fx[f"rs128/{k}/depth"], KW)
    assert maxerr(out[3], fx[f"rs128/{k}/env"]) <= 1e-5
    if n_border == 0:
        assert maxerr(out[2], fx[f"rs128/{k}/bg"]) <= RGB_TOL
    assert out[4].shape == (rays.shape[0], 257)
    kw = dict(n_coarse=512, exp_sampling=True, device=DEV)
    with torch.no_grad():
        out = volume_renderer(rays, model, chunk=4096, **kw)
    n_border = assert_matches_reference(model, rays, out, fx[f"nr512/{k}/rgb"], fx[f"nr512/{k}/depth"], kw)
    if n_border == 0:
        assert maxerr(out[2], fx[f"nr512/{k}/bg"]) <= RGB_TOL


def test_config3_full_image(ricoh, precision):
    """One full 1024 x 2048 render with rays generated on the device: shape, finiteness, the reference's values at the golden
    subset, bit-reproducibility, and row-shard concatenation == the single-process image (config 5's partitioning)."""
    fx, _, model = ricoh
    H, W = int(fx["H"]), int(fx["W"])
    k = 1
    kw = dict(chunk=65536, keep_alpha=False, **KW)
    with torch.no_grad():
        rays = erp_rays(H, W, fx["poses"][k], DEV)
        rgb, depth, bg, env, alpha = volume_renderer(rays, model, **kw)
        assert rgb.shape == (H * W, 3) and depth.shape == (H * W,) and alpha is None
        assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
        assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
        idx = T(fx["idx"])
        assert maxerr(rays[idx], fx[f"rays/{k}"]) <= 1e-6
        assert_matches_reference(model, T(fx[f"rays/{k}"]), (rgb[idx], depth[idx]), fx[f"rs128/{k}/rgb"], fx[f"rs128/{k}/depth"], KW)
        again = volume_renderer(rays, model, **kw)
        assert torch.equal(again[0], rgb) and torch.equal(again[1], depth)
        parts = []
        for r in range(3):  # three ranks' row blocks
            lo, hi = shard_bounds(H, 3, r)
            parts.append(volume_renderer(erp_rays(H, W, fx["poses"][k], DEV, lo, hi - lo), model, **kw)[0])
        assert torch.equal(torch.cat(parts), rgb)


@pytest.mark.parametrize("h", [1000, 1920])
def test_envmap_radiance_at_shipped_sizes(golden, h):
    """models/envmap.py:26-34 at h = 1000 / 1920 on a white-noise map: every texel is independent (range +-3 before the sigmoid),
    so an index slip shows as an O(0.1 .. 1) error, while the float32 rounding of (u, v) itself (1 ulp = 6e-8 of the map =
    1e-4 texel at 3840 texels, times a texel-to-texel step of up to 6, times sigmoid' <= 1/4) already moves values by ~2e-4
    between two correct libms.  Hence: max error <= 2e-3, mean error <= 5e-5; the smooth-map goldens (env in ricoh.npz,
    tiny_envmap.npz) hold to 1e-5."""
    from egonerf_amd.model import EnvironmentMap
    fx = golden("envmap_full")
    env = EnvironmentMap(h=4, init_strategy="zero", device=DEV)
    env.load_envmap(synth.white_envmap(int(fx["seed"]), h), device=DEV)
    with torch.no_grad():
        got = env.get_radiance(T(fx["dirs"]))
    err = np.abs(got.cpu().numpy().astype(np.float64) - fx[f"radiance/{h}"])
    assert err.max() <= 2e-3 and err.mean() <= 5e-5, (err.max(), err.mean())
    assert err[:4].max() <= 1e-6   # +-z and +-x: (u, v) are exact binary fractions there, no rounding slack


def test_alpha_mask_rule_on_the_ricoh_scene(ricoh):
    """Row M at configs[2]'s size (so far only on the tiny / barbershop grids): the reference's mask rule
    (updateAlphaMask: lattice alpha, 3^3 max-pool, alphaMask_thres) + rayMarch_weight_thres on the Ricoh scene, 128+128 with the
    envmap, HIP vs the oracle (whose skip logic is pinned to TensorBase.forward) on the reference's own ERP rays of view 0 -
    poles, seam and region borders included.  Rays holding a weight within rounding of the threshold (a flip moves a colour by
    up to the threshold) or a sample on a yin/yang border (libm-dependent grid choice) are compared for depth only."""
    fx, cfg, model = ricoh
    model.mlp_precision = "f16f6"
    orc = model.oracle
    rays = T(fx["rays/0"])
    kw = {k: v for k, v in KW.items() if k != "device"}
    okw = {k: v for k, v in kw.items() if k != "exp_sampling"}
    thres = 1e-4  # opt.py: rm_weight_mask_thre
    try:
        with torch.no_grad():
            base = model(rays, **kw)
            frac = model.updateAlphaMask()
            assert 0.0 < frac <= 1.0
            vols = [model.alphaMask.alpha_volume_yin.clone(), model.alphaMask.alpha_volume_yang.clone()]
            for v in vols:      # the mask of this smooth field is nearly full: carve two shells and a wedge out so that it bites
                v[..., 40:44] = 0
                v[0, 0, :60, :, 80:] = 0
            from egonerf_amd.model import YinYangAlphaGridMask
            model.alphaMask = YinYangAlphaGridMask(DEV, vols[0], vols[1])
            model.use_alpha_mask, model.use_weight_thres, model.rayMarch_weight_thres = True, True, thres
            model._scene_cache = None
            got = model(rays, **kw)
        orc.alpha_mask, orc.weight_thres = (vols[0].cpu(), vols[1].cpu()), thres
        ref, inter = orc.forward(rays.cpu(), keep=True, **okw)
        # this scene is very transparent (density_shift -10 over 300 units): the weights of a ray pass through the threshold slowly,
        # so many rays hold a sample within rounding of it; each such sample may flip between two fp32 evaluations and then moves
        # the colour by at most its weight (~ the threshold).  Per-ray tolerance: 1e-4 + (samples in the band) x threshold
        n_flip = ((inter["weight"] - thres).abs() < 2e-6).sum(-1)
        pts = [inter["xyz_coarse"]] + ([inter["xyz_fine"]] if "xyz_fine" in inter else [])
        border = torch.stack([orc.yin_margin(p).abs().amin(1) for p in pts]).amin(0) < BORDER_EPS
        err = (got[0].cpu() - ref[0]).abs().amax(1)
        tol = RGB_TOL + n_flip.float() * (thres + 2e-6)
        ok = ~border
        assert int(ok.sum()) >= rays.shape[0] - 10
        assert bool((err[ok] <= tol[ok]).all()), (float((err - tol)[ok].max()), int(n_flip.max()))
        clean = ok & (n_flip == 0)
        assert int(clean.sum()) >= 200 and float(err[clean].max()) <= RGB_TOL
        assert maxerr(got[1][ok.to(DEV)], ref[1][ok]) <= DEPTH_TOL
        assert maxerr(got[3], ref[3]) <= 1e-5                            # the envmap radiance is untouched by the mask
        assert maxerr(got[0], base[0]) > 1e-2                            # ... and the mask is not a no-op on this scene
    finally:
        orc.alpha_mask, orc.weight_thres = None, None
        model.use_alpha_mask = model.use_weight_thres = False
        model.rayMarch_weight_thres = 1e-4
        model.alphaMask = None
        model._scene_cache = None


def test_config3_occupancy_skipping_on_full_size(ricoh):
    """BASELINE configs[2] as written - "full-res ERP render (2048 x 1024), occupancy-grid empty-space skipping ON" (VERDICT r03 item 2):
    the Ricoh scene with real empty space (synth.carve_empty_space: density exactly 0 outside two radial shells and inside a phi
    wedge), the mask built by the reference's rule from the field itself (updateAlphaMask, EgoNeRF.py:437-489: <= 50 % occupied),
    applied with TensorBase.forward's semantics (tensorBase.py:464-478).  The whole 1024 x 2048 image renders with the mask on and
    off; 1 500 of its rays (poles, seam, region borders included) are compared with the oracle carrying THE SAME mask volumes; the
    march's pass-level skip and the shade's tile skip must actually bite; and the skip is exact given the mask: the masked render
    equals the masked render with tile skipping switched off, bit for bit."""
    fx, cfg, base_model = ricoh
    w = synth.carve_empty_space(synth.make_weights(cfg, seed=int(fx["seed_weights"])), cfg)
    model = make_model(cfg, w, DEV)
    orc = make_oracle(cfg, w)
    H, W = 1024, 2048
    pose = fx["poses"][0]
    kw = dict(KW, keep_alpha=False)
    with torch.no_grad():
        rays = erp_rays(H, W, pose, DEV)
        off = volume_renderer(rays, model, chunk=16384, **kw)
        frac = model.updateAlphaMask()
        assert 0.05 < frac <= 0.5, frac
        model.use_alpha_mask = True
        on = volume_renderer(rays, model, chunk=16384, **kw)
        model.skip_zero_weight_tiles = False
        on_noskip = volume_renderer(rays, model, chunk=16384, **kw)
        model.skip_zero_weight_tiles = True
        assert torch.equal(on[0], on_noskip[0]) and torch.equal(on[1], on_noskip[1])
        # the mask is not a no-op, and it only ever removes density: the unmasked render is the same scene plus the
        # softplus(density_shift) haze of "empty" space
        assert maxerr(on[0], off[0]) > 1e-3
        # parity on a spread of rays incl. the first / last rows (poles) and the seam columns
        pick = torch.cat([torch.linspace(0, H * W - 1, 1200).long(), torch.arange(0, 100), torch.arange(H * W - 100, H * W),
                          torch.arange(0, H * W, W)[:50], torch.arange(W - 1, H * W, W)[:50]]).unique()
        sub = rays[pick.to(DEV)]
        got = volume_renderer(sub, model, chunk=4096, **kw)
        assert torch.equal(got[0], on[0][pick.to(DEV)])                   # ray-independent: a subset renders the same bits
        orc.alpha_mask = (model.alphaMask.alpha_volume_yin.cpu(), model.alphaMask.alpha_volume_yang.cpu())
        ref, inter = orc.forward(sub.cpu(), keep=True, **{k: v for k, v in KW.items() if k not in ("exp_sampling", "device")})
        pts = [inter["xyz_coarse"], inter["xyz_fine"]]
        border = torch.stack([orc.yin_margin(p).abs().amin(1) for p in pts]).amin(0) < BORDER_EPS
        err = (got[0].cpu() - ref[0]).abs().amax(1)
        assert int(border.sum()) <= 10
        assert float(err[~border].max()) <= RGB_TOL, float(err[~border].max())
        assert maxerr(got[1][(~border).to(DEV)], ref[1][~border]) <= DEPTH_TOL
        # the skipping is real: per-chunk flags from the march say how many 32-sample tiles the shade never touches
        from egonerf_amd import _lib
        lib, st, sc = _lib.load(), _lib.stream_handle(), model.scene()
        N, S = 16384, 256
        rc = rays[H // 3 * W: H // 3 * W + N].contiguous()
        z = torch.sort(torch.rand(N, S, device=DEV) * 250 + 0.1, dim=1).values.contiguous()
        wgt, bg, crd = torch.empty(N, S, device=DEV), torch.empty(N, device=DEV), torch.empty(N, S, 4, device=DEV)
        act = torch.zeros(N * S // 32, device=DEV, dtype=torch.uint8)
        _lib.check(lib.ego_march_density(sc, rc.data_ptr(), N, S, z.data_ptr(), None, None, 0.1, 2, None, None, 0, wgt.data_ptr(), bg.data_ptr(),
                                         crd.data_ptr(), None, act.data_ptr(), st), "march")
        torch.cuda.synchronize()
        assert float((act == 0).float().mean()) > 0.4      # most tiles of uniformly spread samples lie in empty space and are skipped
