"""Pins oracle/egonerf_oracle.py against vectors captured from the real reference
(oracle/capture_golden.py -> tests/golden/*.npz).  CPU only."""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from oracle.egonerf_oracle import OracleScene, linearised_exp_grid

T = torch.from_numpy


@pytest.fixture(scope="module")
def tiny(golden):
    fx = golden("tiny")
    cfg = synth.SceneConfig(n_voxel=int(fx["n_voxel"]))
    assert cfg.grid == fx["grid"].tolist() == [10, 10, 30]
    return fx, OracleScene(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))


def close(a, b, tol):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    assert a.shape == b.shape and err <= tol, f"max err {err} > {tol}"


def test_rays_regenerate_bit_exact(tiny):
    fx, _ = tiny
    assert np.array_equal(synth.make_rays(64, seed=int(fx["seed_rays"])), fx["rays"])


def test_stage_by_stage(tiny):
    fx, sc = tiny
    rays = T(fx["rays"])
    xyz, z = sc.sample_ray_exp(rays[:, :3], rays[:, 3:6], 24)
    assert np.array_equal(z.numpy(), fx["st_z"])  # schedule is exact by construction
    close(xyz, fx["st_xyz"], 0)
    c7 = sc.from_cartesian(xyz)
    close(c7, fx["st_c7"], 0)
    c7n = sc.normalize_coord(c7)
    close(c7n, fx["st_c7n"], 0)
    sf = sc.density_feature(c7n)
    close(sf, fx["st_sigma_feat"], 1e-6)
    close(sc.density_feature(c7n, coarse=True), fx["st_sigma_feat_coarse"], 1e-6)
    sigma = sc.feature2density(sf)
    close(sigma, fx["st_sigma"], 1e-6)
    d = torch.cat([z[:, 1:] - z[:, :-1], z[:, -1:] - z[:, -2:-1]], -1)
    a, w, bg = sc.raw2alpha(sigma, d * sc.cfg.distance_scale)
    close(a, fx["st_alpha"], 1e-6), close(w, fx["st_weight"], 1e-6), close(bg, fx["st_bg_weight"], 1e-6)
    af = sc.app_feature(c7n)
    close(af, fx["st_app_feat"], 2e-6)
    vd = rays[:, 3:6].view(-1, 1, 3).expand(xyz.shape).reshape(-1, 3)
    close(sc.mlp_fea(vd, af.reshape(-1, 27)).view(64, 24, 3), fx["st_rgb_samples"], 2e-6)


def test_lookups_with_out_of_range_coords(tiny):
    fx, sc = tiny
    q = T(fx["lk_coords"])
    close(sc.density_feature(q), fx["lk_density"], 1e-6)
    close(sc.density_feature(q, coarse=True), fx["lk_density_coarse"], 1e-6)
    close(sc.app_feature(q), fx["lk_app"], 2e-6)


@pytest.mark.parametrize("tag,kw", [
    ("nr", dict(n_coarse=24, n_fine=0, resampling=False)),
    ("rs", dict(n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)),
    ("rsf", dict(n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=False)),
])
def test_end_to_end_eval(tiny, tag, kw):
    fx, sc = tiny
    rgb, depth, _, _, alpha = sc.forward(T(fx["rays"]), **kw)
    close(rgb, fx[f"e2e_{tag}_rgb"], 1e-6)
    close(depth, fx[f"e2e_{tag}_depth"], 2e-5)
    if tag == "nr":  # per-sample alpha only comparable without resampling
        close(alpha, fx["e2e_nr_alpha"], 1e-6)


def test_end_to_end_train_noise_pinned(tiny):
    fx, sc = tiny
    rgb, depth, *_ = sc.forward(T(fx["rays"]), n_coarse=16, n_fine=16, resampling=True, is_train=True,
                                jitter=T(fx["tr_jitter"]), u=T(fx["tr_u"]))
    close(rgb, fx["tr_rgb"], 1e-6)
    close(depth, fx["tr_depth"], 2e-5)


def test_backward_grads_match_reference_autograd(tiny):
    fx, _ = tiny
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    sc = OracleScene(cfg, synth.make_weights(cfg, seed=1234))
    for v in sc.w.values():
        v.requires_grad_(True)
    sc.update_coarse_sigma_grid()
    rgb, *_ = sc.forward(T(fx["rays"]), n_coarse=16, n_fine=16, resampling=True, is_train=True,
                         jitter=T(fx["tr_jitter"]), u=T(fx["tr_u"]))
    loss = torch.mean((rgb - T(fx["bw_gt"])) ** 2)
    assert abs(loss.item() - float(fx["bw_loss"])) < 1e-6
    loss.backward()
    for k, v in sc.w.items():
        ref = fx["bw_grad/" + k]
        g = np.zeros_like(ref) if v.grad is None else v.grad.numpy()
        scale = max(float(np.abs(ref).max()), 1e-8)
        assert float(np.abs(g - ref).max()) <= 2e-5 * scale + 1e-9, k


def test_envmap_variant(golden):
    fx = golden("tiny_envmap")
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=int(fx["envmap_res_H"]))
    sc = OracleScene(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))
    rays = T(fx["rays"])
    rgb, depth, bg, env, alpha = sc.forward(rays, n_coarse=24)
    close(rgb, fx["rgb"], 1e-6), close(depth, fx["depth"], 2e-5), close(bg, fx["bg"], 1e-6)
    close(env, fx["env"], 1e-6), close(alpha, fx["alpha"], 1e-6)
    assert alpha.shape[1] == 25  # trailing ones column (EgoNeRF.py:587)
    close(sc.envmap_radiance(rays[:, 3:6]), fx["radiance"], 1e-6)


def test_schedules_all_scenes(golden):
    fx = golden("stages")
    for name, (near, far, r0) in dict(indoor=(0.01, 15.0, 0.03), ricoh=(0.1, 300.0, 0.05), mid=(0.01, 50.0, 0.05)).items():
        sc = OracleScene.__new__(OracleScene)
        sc.cfg = synth.SceneConfig(n_voxel=20 ** 3, near=near, far=far, r0=r0)
        sc.r0 = r0
        for S in (32, 64, 128, 256, 512):
            z = sc.cfg.near + sc.sample_schedule(S)
            assert np.array_equal(z.numpy(), fx[f"sched/{name}/{S}"]), (name, S)


@pytest.mark.parametrize("name,nv", [("full", 27_000_000), ("tiny", 20 ** 3)])
def test_r_normalisation_and_borders(golden, name, nv):
    fx = golden("stages")
    cfg = synth.SceneConfig(n_voxel=nv)
    sc = OracleScene(cfg, synth.make_weights(synth.SceneConfig(n_voxel=20 ** 3), seed=3)) if nv == 20 ** 3 else None
    if sc is None:  # full grid: only the coordinate machinery is needed, skip the 94 MB of tables
        sc = OracleScene.__new__(OracleScene)
        tiny = OracleScene(synth.SceneConfig(n_voxel=20 ** 3), synth.make_weights(synth.SceneConfig(n_voxel=20 ** 3), seed=3))
        sc.__dict__.update(tiny.__dict__)
        sc.cfg, sc.grid = cfg, list(cfg.grid)
        ratio = pow(sc.far_r / cfg.r0, 1 / (sc.grid[0] - 1))
        sc.r_lut = linearised_exp_grid(cfg.r0, ratio, sc.grid[0] + 1)
    assert float(sc.far_r) == float(fx[f"normr/{name}/far_r"])
    assert np.array_equal(sc.normalize_r(T(fx[f"normr/{name}/r"])).numpy(), fx[f"normr/{name}/out"])
    c7 = sc.from_cartesian(T(fx[f"cart/{name}/xyz"]))
    assert np.array_equal(c7.numpy(), fx[f"cart/{name}/c7"])
    assert np.array_equal(sc.normalize_coord(c7).numpy(), fx[f"cart/{name}/c7n"])


def test_sample_pdf_and_pe(golden):
    fx = golden("stages")
    bins, w = T(fx["pdf/bins"]), T(fx["pdf/weights"])
    close(OracleScene.sample_pdf(bins, w, 32), fx["pdf/eval32"], 0)
    close(OracleScene.sample_pdf(bins, w, 20, u=T(fx["pdf/u"])), fx["pdf/train20"], 0)
    close(OracleScene.positional_encoding(T(fx["pe/x"]), 2), fx["pe/out"], 0)


def test_full_grid_outputs(golden):
    fx = golden("full")
    cfg = synth.SceneConfig()
    assert cfg.grid == fx["grid"].tolist() == [150, 172, 516]
    sc = OracleScene(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))
    rays = T(synth.make_rays(int(fx["n_rays"]), seed=int(fx["seed_rays"])))
    rgb, depth, *_ = sc.forward(rays, n_coarse=64)
    close(rgb, fx["nr64_rgb"], 2e-6), close(depth, fx["nr64_depth"], 5e-5)
    rgb, depth, *_ = sc.forward(rays, n_coarse=32, n_fine=32, resampling=True)
    close(rgb, fx["rs32_rgb"], 5e-6), close(depth, fx["rs32_depth"], 2e-4)
    rgb, depth, *_ = sc.forward(rays[:64], n_coarse=512)
    close(rgb, fx["nr512_rgb"], 5e-6), close(depth, fx["nr512_depth"], 2e-4)


# ---- training-step extras (tests/golden/train_extras.npz, captured from the reference by oracle/capture_golden.py) -------
@pytest.fixture(scope="module")
def extras(golden):
    fx = golden("train_extras")
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=int(fx["envmap_res_H"]))
    return fx, cfg


def _grad_scene(fx, cfg):
    sc = OracleScene(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))
    for v in sc.w.values():
        v.requires_grad_(True)
    sc.update_coarse_sigma_grid()
    return sc


def _check_grads(sc, fx, prefix, rel=2e-5):
    for k, v in sc.w.items():
        ref = fx[f"{prefix}/{k}"]
        g = np.zeros_like(ref) if v.grad is None else v.grad.numpy()
        scale = max(float(np.abs(ref).max()), 1e-8)
        assert float(np.abs(g - ref).max()) <= rel * scale + 1e-9, (prefix, k)


def test_entropy_and_envmap_gradients(extras):
    from oracle.egonerf_oracle import ray_entropy_loss
    fx, cfg = extras
    kw = dict(n_coarse=16, n_fine=16, resampling=True, is_train=True, jitter=T(fx["jitter"]), u=T(fx["u"]))
    sc = _grad_scene(fx, cfg)
    rgb, _, _, _, alpha = sc.forward(T(fx["rays"]), **kw)
    close(rgb, fx["ent_rgb"], 1e-6), close(alpha, fx["ent_alpha"], 1e-6)
    mse, ent = torch.mean((rgb - T(fx["gt"])) ** 2), ray_entropy_loss(alpha)
    assert abs(mse.item() - float(fx["ent_mse"])) < 1e-6 and abs(ent.item() - float(fx["ent_entropy"])) < 1e-5
    (mse + float(fx["entropy_weight"]) * ent).backward()
    _check_grads(sc, fx, "ent_grad")
    assert np.abs(fx["ent_grad/envmap.emission"]).max() > 0
    sc = _grad_scene(fx, cfg)
    ray_entropy_loss(sc.forward(T(fx["rays"]), **kw)[4]).backward()
    _check_grads(sc, fx, "entonly_grad")


def test_envmap_pretraining_gradient(extras):
    fx, cfg = extras
    sc = _grad_scene(fx, cfg)
    env = sc.envmap_radiance(T(fx["rays"])[:, 3:6])
    close(env, fx["pre_env"], 1e-6)
    loss = torch.mean((env - T(fx["gt"])) ** 2)
    assert abs(loss.item() - float(fx["pre_loss"])) < 1e-6
    loss.backward()
    close(sc.w["envmap.emission"].grad, fx["pre_grad"], 1e-8)


@pytest.mark.parametrize("name", ["tv_density", "tv_app", "l1", "ortho"])
def test_regularisers(extras, name):
    fx, cfg = extras
    sc = _grad_scene(fx, cfg)
    v = dict(tv_density=lambda: sc.TV_loss("density"), tv_app=lambda: sc.TV_loss("app"), l1=sc.density_L1,
             ortho=sc.vector_comp_diffs)[name]()
    ref = float(fx[f"reg/{name}/value"])
    assert abs(v.item() - ref) <= 2e-6 * max(abs(ref), 1.0)
    v.backward()
    keys = [k[len(f"reg/{name}/grad/"):] for k in fx.files if k.startswith(f"reg/{name}/grad/")]
    assert keys
    for k in keys:
        g, r = sc.w[k].grad.numpy(), fx[f"reg/{name}/grad/{k}"]
        assert float(np.abs(g - r).max()) <= 2e-5 * max(float(np.abs(r).max()), 1e-8), (name, k)
    assert all(v.grad is None for k, v in sc.w.items() if k not in keys)


def test_upsample_volume_grid(extras):
    fx, cfg = extras
    sc = OracleScene(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))
    sc.upsample_volume_grid(fx["up_target"].tolist())
    sc.set_resolution(fx["up_target"].tolist())  # train.py:376-377: no r0 argument -> the knee resets to 0.05
    assert sc.r0 == 0.05
    for k in [k[3:] for k in fx.files if k.startswith("up/")]:
        assert tuple(sc.w[k].shape) == fx["up/" + k].shape, k
        close(sc.w[k], fx["up/" + k], 5e-6)  # angular axes: linspace + grid_sample vs F.interpolate rounding
    rgb, depth, *_ = sc.forward(T(fx["rays"]), n_coarse=16, n_fine=16, resampling=True)
    close(rgb, fx["up_rgb"], 2e-6), close(depth, fx["up_depth"], 4e-5)


def test_rgb_ssim_restatement(golden):
    from oracle.egonerf_oracle import rgb_ssim, psnr
    fx = golden("metrics")
    a, b = fx["img0"], fx["img1"]
    assert abs(rgb_ssim(a, b, 1) - float(fx["ssim"])) < 1e-12
    assert float(np.abs(rgb_ssim(a, b, 1, return_map=True) - fx["ssim_map"]).max()) < 1e-11
    assert abs(rgb_ssim(a, a, 1) - 1.0) < 1e-12 and abs(float(fx["ssim_same"]) - 1.0) < 1e-12
    assert abs(rgb_ssim(np.full_like(a, 0.25), b, 1) - float(fx["ssim_flat"])) < 1e-12
    assert abs(rgb_ssim(a, b, 1, filter_size=7, filter_sigma=1.0) - float(fx["ssim_fs7"])) < 1e-12
    assert abs(psnr(torch.from_numpy(b), torch.from_numpy(a)) - float(fx["psnr"])) < 1e-9


def test_ws_weights_and_weighted_ssim_vs_reference(golden):
    """extra/ws_ssim.py:12-33 (captured through the real file's estws): oracle restatement and the product's host weights."""
    from egonerf_amd.metrics import weighted_map_mean, ws_weights
    from oracle.egonerf_oracle import ws_mean, ws_psnr, ws_rows
    fx = golden("ws_metrics")
    for n in (7, 64, 1024):
        assert np.array_equal(ws_rows(n), fx[f"ws_rows/{n}"]) and np.array_equal(ws_weights(n), fx[f"ws_rows/{n}"])
    assert np.array_equal(np.repeat(ws_rows(30)[:, None], 46, 1), fx["ws_30x46"])
    assert abs(ws_mean(fx["ssim_map_mean"]) - float(fx["wsssim"])) <= 1e-15
    assert abs(weighted_map_mean(torch.from_numpy(fx["ssim_map_mean"]), ws_weights(30)) - float(fx["wsssim"])) <= 1e-14
    assert np.array_equal(ws_weights(5, 3, 16), ws_rows(16)[3:8])   # a window of rows of a taller panorama
    # constant weights would give the plain PSNR: check the weighted one against a direct double-loop evaluation
    mx = golden("metrics")
    a, b = mx["img0"].astype(np.float64), mx["img1"].astype(np.float64)
    H = a.shape[0]
    num = sum(np.cos((i + 0.5 - H / 2) * np.pi / H) * ((a[i] - b[i]) ** 2).sum() for i in range(H))
    den = sum(np.cos((i + 0.5 - H / 2) * np.pi / H) for i in range(H)) * a.shape[1] * 3
    assert abs(ws_psnr(mx["img1"], mx["img0"]) - 10 * np.log10(den / num)) <= 1e-10


@pytest.mark.parametrize("tag", ["uni", "exp"])
def test_skip_semantics_vs_tensorbase_forward(golden, tag):
    """Row M: the oracle's restatement of TensorBase.forward's mask application + rayMarch_weight_thres appearance skip
    (tensorBase.py:464-507) against the reference's own forward run through TensorVMSplit with an AlphaGridMask
    (tests/golden/skip_semantics.npz), with and without the mask."""
    import torch.nn.functional as F
    from oracle.egonerf_oracle import tensorbase_skip_composite
    fx = golden("skip_semantics")
    z = T(fx[f"{tag}/z"])
    dists = torch.cat([z[:, 1:] - z[:, :-1], z[:, -1:] - z[:, -2:-1]], -1)
    sigma = F.softplus(T(fx[f"{tag}/sigma_feat"]) + float(fx["density_shift"]))
    rays = T(fx["rays"])
    valid, mask_alpha = T(fx[f"{tag}/ray_valid"]), T(fx[f"{tag}/mask_alpha"])
    assert 0.05 < float((valid & (mask_alpha <= 0)).float().mean()) < 0.5          # the mask really removes samples
    for pre, ma in (("", mask_alpha), ("nomask_", None)):
        rgb, depth, alpha, w, _, app = tensorbase_skip_composite(sigma, ma, valid, dists, z, T(fx[f"{tag}/rgb_dense"]), rays[:, -1],
                                                                 float(fx["distance_scale"]), float(fx["weight_thres"]))
        close(rgb.clamp(0, 1), fx[f"{tag}/{pre}rgb"], 1e-6)
        close(depth, fx[f"{tag}/{pre}depth"], 2e-6)
        close(alpha, fx[f"{tag}/{pre}alpha"], 1e-6)
        assert 0.02 < float(((w > 0) & ~app).float().mean())                      # ... and so does the weight threshold
    # the threshold matters: without it the colours move by more than the parity tolerance
    rgb_all, *_ = tensorbase_skip_composite(sigma, mask_alpha, valid, dists, z, T(fx[f"{tag}/rgb_dense"]), rays[:, -1],
                                            float(fx["distance_scale"]), None)
    assert float((rgb_all.clamp(0, 1) - T(fx[f"{tag}/rgb"])).abs().max()) > 1e-3
