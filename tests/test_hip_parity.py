"""GPU parity tests: every call goes through the C ABI of libegonerf_hip.so (via egonerf_amd's host layer)
and is compared with (a) golden vectors captured from the real reference and (b) the CPU oracle.

Tolerances (north_star): RGB 1e-4 absolute, depth 1e-3 * max(z).  Per-stage tolerances are tighter and
written next to each check.  Per-sample alpha is only compared without resampling."""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from tests import mfma_emulator as em
from tests.helpers import make_model, make_oracle, maxerr

pytestmark = pytest.mark.gpu
DEV = "cuda"
RGB_TOL = 1e-4


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def tiny(golden):
    fx = golden("tiny")
    cfg = synth.SceneConfig(n_voxel=int(fx["n_voxel"]))
    w = synth.make_weights(cfg, seed=int(fx["seed_weights"]))
    return fx, cfg, w, make_model(cfg, w, DEV)


@pytest.fixture(scope="module")
def full(golden):
    fx = golden("full")
    cfg = synth.SceneConfig()
    w = synth.make_weights(cfg, seed=int(fx["seed_weights"]))
    return fx, cfg, w, make_model(cfg, w, DEV)


@pytest.fixture(params=["f16x3", "f16f8", "f16f6", "f32"], autouse=True)
def precision(request, tiny, full):
    """Every test runs with all three matrix-product arithmetics (fp16-split MFMA, fp16 + fp8 corrections, fp32 MFMA)."""
    tiny[3].mlp_precision = request.param
    full[3].mlp_precision = request.param
    return request.param


def test_native_library_is_the_path(tiny):
    from egonerf_amd import _lib
    assert _lib.load().ego_abi_version() == _lib.EXPECTED_ABI_VERSION
    with pytest.raises(RuntimeError):  # CPU tensors never silently fall back
        tiny[3](torch.zeros(4, 6), n_coarse=8, exp_sampling=True)


def test_device_packer_is_bit_exact(tiny):
    _, _, w, model = tiny
    model.scene()
    torch.cuda.synchronize()
    blob = model._packed[False].cpu().numpy()   # the inference blob (training packs go to one of their own)
    assert np.array_equal(blob[:em.PACKED_FLOATS], em.pack_mlp(w))
    assert np.array_equal(blob[em.PACKED_FLOATS:2 * em.PACKED_FLOATS].view(np.uint32), em.pack_mlp_f16(w).view(np.uint32))
    assert blob.shape[0] == 2 * em.PACKED_FLOATS + 9216 + em.F8_FLOATS + em.F6_FLOATS  # + basis fragments in the fp16-table K order + f16f8 + f16f6 images
    f8 = 2 * em.PACKED_FLOATS + 9216
    assert np.array_equal(blob[f8:f8 + em.F8_FLOATS].view(np.uint32), em.pack_mlp_f8(w).view(np.uint32))
    assert np.array_equal(blob[f8 + em.F8_FLOATS:].view(np.uint32), em.pack_mlp_f6(w).view(np.uint32))


def test_stage_sample_and_coords(tiny):
    fx, cfg, _, model = tiny
    rays = T(fx["rays"])
    xyz, z, _ = model.sample_ray_exp(rays[:, :3], rays[:, 3:6], is_train=False, N_samples=24)
    assert np.array_equal(z.cpu().numpy(), fx["st_z"])
    assert maxerr(xyz, fx["st_xyz"]) == 0.0
    c7 = model.coordinates.from_cartesian(T(fx["st_xyz"]))
    assert np.array_equal(c7[..., 6].cpu().numpy(), fx["st_c7"][..., 6])  # same grid choice for every sample
    assert maxerr(c7, fx["st_c7"]) <= 2e-6  # acos/atan2: ocml vs sleef, ~1-2 ulp of pi
    c7n = model.coordinates.normalize_coord(T(fx["st_c7"]), downsample=2)
    assert maxerr(c7n, fx["st_c7n"]) <= 2.4e-7  # identical op sequence; division/rounding ties only


def test_stage_coords_edge_cases(golden, tiny):
    fx = golden("stages")
    model = tiny[3]
    for name in ("tiny",):
        xyz = fx[f"cart/{name}/xyz"]
        c7 = model.coordinates.from_cartesian(T(xyz)).cpu().numpy()
        ref = fx[f"cart/{name}/c7"]
        # points exactly on a yin/yang border (theta = pi/4 ... ) may pick either grid: acos/atan2 differ by an
        # ulp between ocml and the CPU's sleef.  Everything farther than 1e-5 rad from a border must agree.
        x, y, z_ = xyz.astype(np.float64).T
        with np.errstate(invalid="ignore", divide="ignore"):
            th = np.nan_to_num(np.arccos(z_ / np.sqrt(x * x + y * y + z_ * z_)))
        ph = np.arctan2(y, x)
        margin = np.minimum.reduce([np.abs(th - np.pi / 4), np.abs(th - 3 * np.pi / 4), np.abs(ph - 3 * np.pi / 4),
                                    np.abs(ph + 3 * np.pi / 4)])
        safe = margin > 1e-5
        assert safe.sum() >= len(xyz) - 6 and not safe[10]  # (1,0,1) sits exactly on theta = pi/4
        assert np.array_equal(c7[safe, 6], ref[safe, 6])  # incl. r = 0 and the poles
        assert float(np.abs(c7[safe] - ref[safe]).max()) <= 4e-6
        assert maxerr(model.coordinates.normalize_coord(T(ref)), fx[f"cart/{name}/c7n"]) <= 5e-7
        r = fx[f"normr/{name}/r"]
        c = np.zeros((r.shape[0], 7), np.float32)
        c[:, 0] = r
        out = model.coordinates.normalize_coord(T(c))[:, 0].cpu().numpy()
        assert np.max(np.abs(out - (fx[f"normr/{name}/out"] * 2 - 1))) <= 2.4e-7


def test_stage_lookups(tiny):
    fx, cfg, _, model = tiny
    c7n = T(fx["st_c7n"])
    assert maxerr(model.compute_densityfeature(c7n), fx["st_sigma_feat"]) <= 2e-6
    assert maxerr(model.compute_coarse_densityfeature(c7n), fx["st_sigma_feat_coarse"]) <= 2e-6
    assert maxerr(model.compute_appfeature(c7n), fx["st_app_feat"]) <= 4e-6
    q = T(fx["lk_coords"])  # out-of-range coordinates: zero padding
    assert maxerr(model.compute_densityfeature(q), fx["lk_density"]) <= 2e-6
    assert maxerr(model.compute_coarse_densityfeature(q), fx["lk_density_coarse"]) <= 2e-6
    assert maxerr(model.compute_appfeature(q), fx["lk_app"]) <= 4e-6
    assert model.compute_densityfeature(q[:0]).shape == (0,)  # empty input


def test_stage_density_alpha_mlp(tiny, precision):
    from egonerf_amd.model import raw2alpha
    fx, cfg, _, model = tiny
    sigma = model.feature2density(T(fx["st_sigma_feat"]))
    assert maxerr(sigma, fx["st_sigma"]) <= 1e-6
    z = T(fx["st_z"])
    d = torch.cat([z[:, 1:] - z[:, :-1], z[:, -1:] - z[:, -2:-1]], -1)
    a, w, bg = raw2alpha(T(fx["st_sigma"]), d * cfg.distance_scale)
    assert maxerr(a, fx["st_alpha"]) <= 1e-6 and maxerr(w, fx["st_weight"]) <= 1e-6 and maxerr(bg, fx["st_bg_weight"]) <= 1e-6
    rays = T(fx["rays"])
    vd = rays[:, 3:6].view(-1, 1, 3).expand(64, 24, 3)
    rgb = model.renderModule(None, vd, T(fx["st_app_feat"]))
    # per-sample colour: fp32-grade for the three-term arithmetics; with fp8 correction terms a single sample may be off by
    # a few 1e-5 (the composited colour, which is what the 1e-4 bar applies to, by < 1e-5: see the end-to-end tests)
    assert maxerr(rgb, fx["st_rgb_samples"]) <= (5e-5 if precision in ("f16f8", "f16f6") else 5e-6)


def test_stage_sample_pdf(golden):
    """Inverse-CDF + merge against the oracle's sample_pdf (itself pinned to the reference by stages.npz).

    The reference algorithm is discontinuous where a cdf step is ~1e-5 (`denom < 1e-5 -> 1`, ray_utils.py:182)
    and where u ties a cdf value, so exact agreement is asserted on well-conditioned weights (plus an all-zero
    row = uniform pdf); on the golden weights (u^4: many ~0 bins) only the non-degenerate entries must agree."""
    from egonerf_amd import _lib
    from oracle.egonerf_oracle import OracleScene
    fx = golden("stages")
    z = torch.sort(torch.from_numpy((synth.hash_uniform(31, 0, 8 * 32).reshape(8, 32) * 12).astype(np.float32)), -1)[0]
    mids = 0.5 * (z[:, 1:] + z[:, :-1])
    lib = _lib.load()
    zt = z.to(DEV)
    w_good = torch.zeros(8, 32)
    w_good[:, 1:-1] = torch.from_numpy((0.05 + 0.95 * synth.hash_uniform(32, 0, 8 * 30).reshape(8, 30)).astype(np.float32))
    w_good[3] = 0
    w_ill = torch.zeros(8, 32)
    w_ill[:, 1:-1] = torch.from_numpy(fx["pdf/weights"])
    for w, exact in ((w_good, True), (w_ill, False)):
        wt = w.to(DEV)
        for n, u in ((32, None), (20, torch.from_numpy(fx["pdf/u"]))):
            expect = OracleScene.sample_pdf(mids, w[:, 1:-1], n, u)
            for use_coarse in (1, 0):
                z_out = torch.empty(8, (32 if use_coarse else 0) + n, device=DEV)
                z_new = torch.empty(8, n, device=DEV)
                _lib.check(lib.ego_sample_pdf_merge(zt.data_ptr(), wt.data_ptr(), _lib.ptr(None if u is None else u.to(DEV)), 8, 32,
                                                    n, use_coarse, z_out.data_ptr(), z_new.data_ptr(), _lib.stream_handle()),
                           "ego_sample_pdf_merge")
                err = (z_new.cpu() - expect).abs()
                if exact:
                    assert float(err.max()) <= 1e-5
                else:
                    assert float((err <= 1e-5).float().mean()) >= 0.95
                    # ... and every entry that differs is explained by the reference algorithm's own conditioning, evaluated in
                    # float64: (a) the `denom < 1e-5 -> 1` branch (ray_utils.py:182) within rounding of its threshold, (b) u within
                    # rounding of a cdf knot (searchsorted picks the neighbouring bin), or (c) a thin cdf step, where
                    # (u - cdf_lo) / denom amplifies the ~1e-7 rounding of the float32 cdf to 4e-7 / denom of a bin width
                    w64 = w[:, 1:-1].double() + 1e-5
                    cdf = torch.cat([torch.zeros(8, 1, dtype=torch.float64), torch.cumsum(w64 / w64.sum(-1, keepdim=True), -1)], -1)
                    uu = (torch.linspace(0.0, 1.0, n).expand(8, n) if u is None else u).double().contiguous()
                    idx = torch.searchsorted(cdf, uu, right=True)
                    lo, hi = (idx - 1).clamp(min=0), idx.clamp(max=cdf.shape[-1] - 1)
                    denom = torch.gather(cdf, -1, hi) - torch.gather(cdf, -1, lo)
                    width = (torch.gather(mids.double(), -1, hi) - torch.gather(mids.double(), -1, lo)).abs()
                    knot_gap = (cdf[:, None, :] - uu[:, :, None]).abs().amin(-1)
                    span = float(mids.max() - mids.min())
                    explained = ((denom - 1e-5).abs() < 2e-6) | (knot_gap < 4e-7) | (err.double() <= 1e-5 + 4e-7 / denom.clamp(min=1e-12) * width) \
                        | ((denom < 1e-5) & (err.double() <= 4e-7 * span + 1e-5))
                    assert bool(explained[err > 1e-5].all()), (err[err > 1e-5], denom[err > 1e-5], knot_gap[err > 1e-5])
                merged = torch.sort(torch.cat([zt, z_new], -1) if use_coarse else z_new, -1)[0]
                assert torch.equal(z_out, merged)  # the sort itself is exact


@pytest.mark.parametrize("tag,kw", [
    ("nr", dict(n_coarse=24, n_fine=0, resampling=False)),
    ("rs", dict(n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)),
    ("rsf", dict(n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=False)),
])
def test_e2e_tiny_vs_reference(tiny, tag, kw):
    from egonerf_amd.renderer import volume_renderer
    fx, cfg, _, model = tiny
    with torch.no_grad():
        rgb, depth, bg, env, alpha = volume_renderer(T(fx["rays"]), model, chunk=4096, exp_sampling=True, device=DEV,
                                                     interval_th=True, **kw)
    assert bg is None and env is None
    assert maxerr(rgb, fx[f"e2e_{tag}_rgb"]) <= RGB_TOL
    assert maxerr(depth, fx[f"e2e_{tag}_depth"]) <= 1e-3 * 15.0
    if tag == "nr":
        assert maxerr(alpha, fx["e2e_nr_alpha"]) <= 1e-5


def test_e2e_tiny_train_noise_pinned(tiny):
    fx, cfg, _, model = tiny
    with torch.no_grad():
        rgb, depth, *_ = model(T(fx["rays"]), is_train=True, n_coarse=16, n_fine=16, exp_sampling=True, resampling=True,
                               use_coarse_sample=True, jitter=T(fx["tr_jitter"]), u=T(fx["tr_u"]))
    assert maxerr(rgb, fx["tr_rgb"]) <= RGB_TOL
    assert maxerr(depth, fx["tr_depth"]) <= 1e-3 * 15.0


def test_e2e_tiny_envmap(golden, precision):
    fx = golden("tiny_envmap")
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=int(fx["envmap_res_H"]))
    w = synth.make_weights(cfg, seed=int(fx["seed_weights"]))
    model = make_model(cfg, w, DEV)
    model.mlp_precision = precision
    rays = T(fx["rays"])
    with torch.no_grad():
        rgb, depth, bg, env, alpha = model(rays, n_coarse=24, exp_sampling=True)
    assert alpha.shape == (64, 25) and torch.all(alpha[:, -1] == 1)  # trailing ones column (EgoNeRF.py:587)
    assert maxerr(rgb, fx["rgb"]) <= RGB_TOL and maxerr(bg, fx["bg"]) <= RGB_TOL and maxerr(env, fx["env"]) <= 1e-5
    assert maxerr(depth, fx["depth"]) <= 1e-3 * 15.0 and maxerr(alpha, fx["alpha"]) <= 1e-5
    assert maxerr(model.envmap.get_radiance(rays[:, 3:6]), fx["radiance"]) <= 1e-5
    assert maxerr(model(rays, pretrain_envmap=True), fx["radiance"]) <= 1e-5


@pytest.mark.parametrize("tag,n,kw", [
    ("nr64", 256, dict(n_coarse=64)),
    ("nr512", 64, dict(n_coarse=512)),
    ("rs32", 256, dict(n_coarse=32, n_fine=32, resampling=True)),
    ("rs128", 64, dict(n_coarse=128, n_fine=128, resampling=True)),
])
def test_e2e_full_grid_vs_reference(full, tag, n, kw):
    fx, cfg, _, model = full
    rays = T(synth.make_rays(int(fx["n_rays"]), seed=int(fx["seed_rays"])))[:n]
    with torch.no_grad():
        rgb, depth, *_ = model(rays, exp_sampling=True, **kw)
    assert maxerr(rgb, fx[f"{tag}_rgb"]) <= RGB_TOL
    assert maxerr(depth, fx[f"{tag}_depth"]) <= 1e-3 * 23.3
    # north_star: "within 1e-4 RGB and 1e-3 PSNR" - the PSNR clause stated directly (renderer.py:156-157): ground truth = the reference's
    # own image + integer-hash noise at ~30 dB; PSNR(HIP, gt) and PSNR(reference, gt) must agree to 1e-3 dB
    d_psnr, p_hip, p_ref = synth.delta_psnr(rgb.cpu().numpy(), fx[f"{tag}_rgb"])
    assert 28.0 < p_ref < 36.0 and abs(d_psnr) <= 1e-3, (tag, d_psnr, p_hip, p_ref)


def test_full_size_config2_properties(full):
    """BASELINE config 2 shape (4096 rays x 512 samples): size-independent properties + oracle spot check."""
    _, cfg, w, model = full
    rays = T(synth.make_rays(4096, seed=1))
    with torch.no_grad():
        a = model(rays, n_coarse=512, exp_sampling=True)
        b = model(rays, n_coarse=512, exp_sampling=True)
        parts = [model(rays[i:i + 1000], n_coarse=512, exp_sampling=True) for i in range(0, 4096, 1000)]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[4], b[4])  # deterministic
    assert torch.equal(torch.cat([p[0] for p in parts]), a[0])  # rays independent: shard-concat is bit-identical
    assert torch.equal(torch.cat([p[1] for p in parts]), a[1])
    assert float(a[0].min()) >= 0.0 and float(a[0].max()) <= 1.0 and bool(torch.isfinite(a[1]).all())
    assert a[4].shape == (4096, 512) and float(a[4].min()) >= 0.0 and float(a[4].max()) <= 1.0
    oracle = make_oracle(cfg, w)
    sel = slice(0, 96)
    ref = oracle.forward(rays[sel].cpu(), n_coarse=512)
    assert maxerr(a[0][sel], ref[0]) <= RGB_TOL
    assert maxerr(a[1][sel], ref[1]) <= 1e-3 * 23.3
    assert maxerr(a[4][sel], ref[4]) <= 2e-5
    mse = float(((a[0][sel].cpu() - ref[0]) ** 2).mean())
    assert -10 * np.log10(max(mse, 1e-30)) >= 80.0  # PSNR(build vs oracle) >= 80 dB


def test_abi_rejects_bad_arguments(tiny):
    from egonerf_amd import _lib
    lib = _lib.load()
    model = tiny[3]
    sc = model.scene()
    assert lib.ego_shade(sc, None, None, None, 4, 8, None, None, None, None) == -1 and b"null" in lib.ego_last_error()
    bad = _lib.Scene.from_buffer_copy(sc)
    bad.app_dim = 40  # outside what any kernel of the library covers (tuned 27; compatibility kernels 1..32)
    x = torch.zeros(8, 7, device=DEV)
    out = torch.zeros(8, 27, device=DEV)
    assert lib.ego_app_feature(bad, x.data_ptr(), 8, out.data_ptr(), None) == -2  # EGO_E_UNSUPPORTED
    assert b"app_dim" in lib.ego_last_error()


@pytest.mark.parametrize("n_rays,kw", [
    (37, dict(n_coarse=70)),                                            # S not a multiple of 32/64, N not of 4
    (5, dict(n_coarse=33, n_fine=17, resampling=True)),                  # odd coarse/fine counts
    (1, dict(n_coarse=3, n_fine=2, resampling=True, use_coarse_sample=False)),  # minimum sizes the sampler allows
    (130, dict(n_coarse=2)),                                             # two samples per ray
])
def test_ragged_sizes_vs_oracle(tiny, n_rays, kw):
    _, cfg, w, model = tiny
    oracle = make_oracle(cfg, w)
    rays = torch.from_numpy(synth.make_rays(n_rays, seed=21))
    with torch.no_grad():
        rgb, depth, _, _, alpha = model(rays.to(DEV), exp_sampling=True, **kw)
    ref = oracle.forward(rays, **kw)
    assert rgb.shape == (n_rays, 3) and alpha.shape == ref[4].shape
    assert maxerr(rgb, ref[0]) <= RGB_TOL and maxerr(depth, ref[1]) <= 1e-3 * 15.0
    if not kw.get("resampling"):
        assert maxerr(alpha, ref[4]) <= 1e-5


def test_empty_batch(tiny):
    model = tiny[3]
    with torch.no_grad():
        rgb, depth, _, _, alpha = model(torch.zeros(0, 6, device=DEV), n_coarse=16, exp_sampling=True)
    assert rgb.shape == (0, 3) and depth.shape == (0,) and alpha.shape == (0, 16)


def test_alpha_mask_matches_reference(golden, tiny):
    """Row M: updateAlphaMask / sample_alpha against the reference's own run (tests/golden/alpha_mask.npz)."""
    fx = golden("alpha_mask")
    _, cfg, w, model = tiny
    assert abs(float(model.stepSize) - float(fx["step_size"])) <= 1e-7
    frac = model.updateAlphaMask(tuple(cfg.grid))
    am = model.alphaMask
    for name, vol in (("vol_yin", am.alpha_volume_yin), ("vol_yang", am.alpha_volume_yang)):
        got = vol.cpu().numpy().astype(np.uint8)
        assert got.shape == fx[name].shape
        assert float((got != fx[name]).mean()) <= 2e-3  # voxels whose pooled alpha sits within rounding of the threshold
    # sample_alpha on the reference's volumes (so the lookup is compared on identical inputs)
    from egonerf_amd.model import YinYangAlphaGridMask
    ref_mask = YinYangAlphaGridMask(DEV, T(fx["vol_yin"].astype(np.float32)), T(fx["vol_yang"].astype(np.float32)))
    assert maxerr(ref_mask.sample_alpha(T(fx["coords"])), fx["sampled"]) <= 2e-6
    assert 0.0 < frac <= 1.0
    model.alphaMask = None
    model._scene_cache = None


def test_skip_semantics_on_reference_mask(golden, tiny):
    """Row M end to end: the mask volumes the REFERENCE built (tests/golden/alpha_mask.npz: its updateAlphaMask on this scene)
    + rayMarch_weight_thres, applied by the HIP path and by the oracle, whose skip logic is pinned to the reference's
    TensorBase.forward (tests/test_oracle_golden.py::test_skip_semantics_vs_tensorbase_forward)."""
    from egonerf_amd.model import YinYangAlphaGridMask
    fx = golden("alpha_mask")
    _, cfg, w, model = tiny
    oracle = make_oracle(cfg, w)
    rays = torch.from_numpy(synth.make_rays(300, seed=31))
    vols = [fx["vol_yin"].astype(np.float32), fx["vol_yang"].astype(np.float32)]
    # the reference's mask of this smooth synthetic field is almost full; carve two radial shells and a wedge out of it so
    # the mask really removes samples (still {0,1} volumes (1,1,N_phi,N_theta,N_r) as the reference stores them)
    for v in vols:
        v[..., 3:5] = 0
        v[0, 0, :8, :, 6:] = 0
    try:
        model.alphaMask = YinYangAlphaGridMask(DEV, T(vols[0]), T(vols[1]))
        oracle.alpha_mask = (torch.from_numpy(vols[0]), torch.from_numpy(vols[1]))
        for thres, kw in ((1e-4, dict(n_coarse=64)), (2e-2, dict(n_coarse=64)), (2e-2, dict(n_coarse=32, n_fine=32, resampling=True)),
                          (5e-2, dict(n_coarse=40))):
            model.use_alpha_mask, model.use_weight_thres, model.rayMarch_weight_thres = True, True, thres
            oracle.weight_thres = thres
            with torch.no_grad():
                got = model(rays.to(DEV), exp_sampling=True, **kw)
            ref, inter = oracle.forward(rays, keep=True, **kw)
            # a weight within rounding of the threshold flips between two fp32 evaluations and moves a colour by up to
            # `thres`: rays holding such a sample are compared for depth / alpha only
            ok = ~((inter["weight"] - thres).abs() < 2e-6).any(-1)
            assert int(ok.sum()) >= 290
            assert maxerr(got[0][ok.to(DEV)], ref[0][ok]) <= RGB_TOL and maxerr(got[1], ref[1]) <= 1e-3 * 15.0
            if not kw.get("resampling"):
                assert maxerr(got[4], ref[4]) <= 2e-5
                skipped = ((inter["weight"] <= thres) & (inter["weight"] > 0)).float().mean()
                assert float(skipped) > 0.02
        # the options change the image (they are not no-ops) ...
        model.use_alpha_mask = model.use_weight_thres = False
        oracle.alpha_mask, oracle.weight_thres = None, None
        with torch.no_grad():
            unskipped = model(rays.to(DEV), exp_sampling=True, n_coarse=40)
        assert maxerr(unskipped[0], oracle.forward(rays, n_coarse=40)[0]) <= RGB_TOL
        assert maxerr(unskipped[0], got[0]) > 1e-2
    finally:
        model.use_alpha_mask = model.use_weight_thres = False
        model.rayMarch_weight_thres = 1e-4
        model.alphaMask = None
        model._scene_cache = None


def test_masked_and_terminated_render_vs_oracle(full):
    """Opt-in skipping on the full-size grid: mask (built here, construction pinned by test_alpha_mask_matches_reference) +
    early termination + the reference's default weight threshold in HIP and oracle; plus the error bound of early
    termination against the unskipped render (|d rgb| <= eps)."""
    _, cfg, w, model = full
    oracle = make_oracle(cfg, w)
    rays = torch.from_numpy(synth.make_rays(96, seed=13))
    kw = dict(n_coarse=256)
    try:
        with torch.no_grad():
            base = model(rays.to(DEV), exp_sampling=True, **kw)
            frac = model.updateAlphaMask()
            model.use_alpha_mask = True
            model.early_termination_eps = 1e-5
            got = model(rays.to(DEV), exp_sampling=True, **kw)
        oracle.alpha_mask = (model.alphaMask.alpha_volume_yin.cpu(), model.alphaMask.alpha_volume_yang.cpu())
        oracle.term_eps = 1e-5
        ref = oracle.forward(rays, **kw)
        assert maxerr(got[0], ref[0]) <= RGB_TOL and maxerr(got[1], ref[1]) <= 1e-3 * 23.3
        assert maxerr(got[4], ref[4]) <= 2e-5
        assert 0.0 < frac < 1.0
        # early termination alone is parity-safe by construction: it drops at most eps of transmittance.  (The reference's
        # mask rule - alpha per grid step >= 1e-4 - is not: on this semi-transparent synthetic field it moves RGB by ~0.1,
        # which is why it stays opt-in, as in EgoNeRF.forward.)
        model.use_alpha_mask = False
        model._scene_cache = None
        with torch.no_grad():
            term = model(rays.to(DEV), exp_sampling=True, **kw)
        assert maxerr(term[0], base[0]) <= 1.5e-5
    finally:
        model.use_alpha_mask = False
        model.early_termination_eps = 0.0
        model.alphaMask = None
        model._scene_cache = None


def test_erp_rays_on_device():
    """ego_erp_rays vs the numpy restatement of get_ray_directions_360/get_rays (egonerf_amd.synth.erp_rays)."""
    from egonerf_amd.renderer import erp_rays
    H, W = 16, 32
    eye = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    got = erp_rays(H, W, eye, DEV).cpu().numpy()
    assert np.abs(got - synth.erp_rays(H, W)).max() <= 3e-7
    # rotated / translated pose, a window of rows
    c, s_ = np.cos(0.3), np.sin(0.3)
    pose = np.array([[c, 0, s_, 0.1], [0, 1, 0, -0.2], [-s_, 0, c, 0.05]], np.float32)
    win = erp_rays(H, W, pose, DEV, row0=4, n_rows=5).cpu().numpy()
    ref = synth.erp_rays(H, W, 4, 9)
    ref = np.concatenate([np.broadcast_to(pose[:, 3], (ref.shape[0], 3)), ref[:, 3:] @ pose[:, :3].T], 1)
    assert np.abs(win - ref).max() <= 5e-7


def test_reference_checkpoint_renders_like_the_reference(golden):
    """Load the reference-written `.th` (shim module paths), save it again in the same format, reload, render."""
    import os, tempfile
    from egonerf_amd.compat import load_reference_checkpoint
    fx = golden("reference_ckpt_render")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_ckpt.th")
    model, step = load_reference_checkpoint(path, device=DEV)
    assert step == 4321
    with tempfile.TemporaryDirectory() as d:
        model.save(os.path.join(d, "again.th"), 99)
        model, step = load_reference_checkpoint(os.path.join(d, "again.th"), device=DEV)
    assert step == 99
    with torch.no_grad():
        rgb, depth, bg, env, alpha = model(T(fx["rays"]), n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)
    assert maxerr(rgb, fx["rgb"]) <= RGB_TOL and maxerr(depth, fx["depth"]) <= 1e-3 * 15.0
    assert maxerr(bg, fx["bg"]) <= RGB_TOL and maxerr(env, fx["env"]) <= 1e-5 and alpha.shape == fx["alpha"].shape


def test_fp16_appearance_tables(full):
    """Optional half-precision shadow of the appearance tables (inference): parity against the oracle on the fp32 tables
    (tolerance 1e-4 like every other path), and against the fp32-table render to bound what the storage format costs."""
    _, cfg, w, model = full
    if model.mlp_precision == "f32":
        pytest.skip("the fp16-table gather exists in the fp16-split kernels")
    rays = torch.from_numpy(synth.make_rays(256, seed=4))
    oracle = make_oracle(cfg, w)
    try:
        with torch.no_grad():
            base = model(rays.to(DEV), n_coarse=128, exp_sampling=True)
            model.app_table_dtype = "f16"
            got = model(rays.to(DEV), n_coarse=128, exp_sampling=True)
            feat16 = model.compute_appfeature(T(np.concatenate([np.random.RandomState(0).uniform(-1, 1, (512, 6)), np.zeros((512, 1))], 1).astype(np.float32)))
        ref = oracle.forward(rays, n_coarse=128)
        assert maxerr(got[0], ref[0]) <= RGB_TOL and maxerr(got[1], ref[1]) <= 1e-3 * 23.3
        assert maxerr(got[0], base[0]) <= 8e-5          # fp16 storage (2^-11 per entry) costs ~5e-5 here: opt-in, not default
        assert torch.equal(got[4], base[4])             # density path untouched
        assert bool(torch.isfinite(feat16).all())
    finally:
        model.app_table_dtype = "f32"


@pytest.mark.parametrize("kw", [dict(n_coarse=512), dict(n_coarse=96, n_fine=96, resampling=True, use_coarse_sample=True), dict(n_coarse=77)])
def test_zero_weight_tile_skip_is_exact(kw, precision):
    """The default-on tile skip (32-sample tiles whose weights are all exactly 0 are not shaded, ego_scene.weight_thres = 0) must
    not change a bit of any output: the reference adds w * rgb = 0 for those samples (EgoNeRF.py:583).  Opaque variant of the
    synthetic field (density_shift 0: the transmittance underflows behind the first surfaces), where most tiles are skipped."""
    cfg = synth.SceneConfig(n_voxel=40 ** 3, density_shift=0.0)
    model = make_model(cfg, synth.make_weights(cfg, seed=77), DEV)
    model.mlp_precision = precision
    rays = T(synth.make_rays(300, seed=5))
    kw = dict(kw, exp_sampling=True)
    assert model.skip_zero_weight_tiles  # the default
    with torch.no_grad():
        on = model(rays, **kw)
        lean = model(rays, need_alpha=False, **kw)   # + the march stops evaluating a ray behind transmittance 0 (exact as well)
        model.skip_zero_weight_tiles = False
        off = model(rays, **kw)
        lean_off = model(rays, need_alpha=False, **kw)
        # what the skip had to work with: weights of the last march, per 32-sample tile of the flat [N * S] order
        xyz, z = model.sample_ray_exp(rays[:, :3], rays[:, 3:], is_train=False, N_samples=kw["n_coarse"])[:2] if "n_fine" not in kw else (None, None)
    for a, b in zip(on, off):
        if a is not None:
            assert torch.equal(a, b)
    assert lean[4] is None and lean_off[4] is None
    for k in range(4):
        if off[k] is not None:
            assert torch.equal(lean[k], off[k]) and torch.equal(lean_off[k], off[k])
    if z is not None:
        S = kw["n_coarse"]
        c7 = model.coordinates.normalize_coord(model.coordinates.from_cartesian(xyz.reshape(-1, 3)))
        sigma = model.feature2density(model.compute_densityfeature(c7)).view(-1, S)
        dist = torch.cat([z[:, 1:] - z[:, :-1], z[:, -1:] - z[:, -2:-1]], -1)
        from egonerf_amd.model import raw2alpha
        w = raw2alpha(sigma, dist * model.distance_scale)[1].reshape(-1)
        pad = (-w.numel()) % 32
        tiles = torch.cat([w, w.new_zeros(pad)]).view(-1, 32)
        frac = float((tiles.max(dim=1).values == 0).float().mean())
        assert frac > (0.3 if S == 512 else 0.0)   # the scene really exercises the skip (tiles straddle rays at S = 77)
