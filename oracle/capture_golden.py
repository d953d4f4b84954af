"""Golden-vector capture — runs ONLY in the build container (needs /root/reference, read-only).

Imports the real reference with stub modules for its missing third-party imports,
feeds it the deterministic synthetic scenes of egonerf_amd/synth.py and writes small .npz fixtures
to tests/golden/.  Neither the reference nor any bytecode of it is copied anywhere; fixtures hold
inputs (or seeds) and expected outputs only.

    python oracle/capture_golden.py            # regenerate every fixture
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np
import torch


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_stub("kornia", create_meshgrid=lambda *a, **k: None)
_stub("cv2", COLORMAP_JET=2)
_tv = _stub("torchvision")
_tv.transforms = _stub("torchvision.transforms")
_stub("imageio")
_stub("plyfile", PlyData=None, PlyElement=None)
_sk = _stub("skimage")
_sk.measure = _stub("skimage.measure")
_stub("lpips")

with contextlib.redirect_stdout(io.StringIO()):
    from models.EgoNeRF import EgoNeRF, YinYangAlphaGridMask  # noqa: E402
    from models.coordinates import YinYangSphericalCoords  # noqa: E402
    from models.envmap import EnvironmentMap  # noqa: E402
    from models.tensorBase import raw2alpha, positional_encoding  # noqa: E402
    from dataLoader.ray_utils import sample_pdf  # noqa: E402
    from renderer import volume_renderer  # noqa: E402
    import sampler as ref_sampler  # noqa: E402

from egonerf_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def build_reference(cfg: synth.SceneConfig, weights):
    aabb = torch.from_numpy(cfg.aabb)
    with contextlib.redirect_stdout(io.StringIO()):
        coords = YinYangSphericalCoords("cpu", aabb, exp_r=True, N_voxel=cfg.n_voxel, r0=cfg.r0, interval_th=cfg.interval_th)
        reso = coords.N_to_reso(cfg.n_voxel, aabb)
        assert reso == cfg.grid, (reso, cfg.grid)
        model = EgoNeRF(aabb, reso, "cpu", coords, density_n_comp=list(cfg.density_n_comp),
                        appearance_n_comp=list(cfg.app_n_comp), app_dim=cfg.app_dim, near_far=[cfg.near, cfg.far],
                        shadingMode=cfg.shadingMode, alphaMask_thres=1e-4, density_shift=cfg.density_shift,
                        distance_scale=cfg.distance_scale, pos_pe=6, view_pe=cfg.view_pe, fea_pe=cfg.fea_pe,
                        featureC=cfg.featureC, step_ratio=0.5, fea2denseAct="softplus", use_envmap=cfg.use_envmap,
                        envmap_res_H=cfg.envmap_res_H, coarse_sigma_grid_update_rule="conv",
                        coarse_sigma_grid_reso=None, interval_th=cfg.interval_th)
    sd = {k: torch.from_numpy(v) for k, v in weights.items() if k != "envmap.emission"}
    model.load_state_dict(sd)
    if cfg.use_envmap:
        model.envmap.load_envmap(weights["envmap.emission"], device="cpu")
    model.update_coarse_sigma_grid()
    model.eval()
    return model, coords


class patched_rand:
    """Make torch.rand_like / torch.rand return queued tensors (pins is_train noise)."""

    def __init__(self, like_queue, rand_queue):
        self.like_queue, self.rand_queue = list(like_queue), list(rand_queue)

    def __enter__(self):
        self._rl, self._r = torch.rand_like, torch.rand
        torch.rand_like = lambda t, **k: self.like_queue.pop(0).to(t.dtype).reshape(t.shape)
        torch.rand = lambda *a, **k: self.rand_queue.pop(0)
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.rand = self._rl, self._r


def run_forward(model, rays, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return volume_renderer(rays, model, chunk=4096, exp_sampling=True, device="cpu", interval_th=True, **kw)


def np_(t):
    return None if t is None else t.detach().cpu().numpy()


def capture_tiny():
    """Tiny grid [10,10,30]: every intermediate of SURVEY 3.3, both resampling settings, train noise, envmap."""
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    weights = synth.make_weights(cfg, seed=1234)
    model, coords = build_reference(cfg, weights)
    rays = torch.from_numpy(synth.make_rays(64, seed=7))
    fx = dict(seed_weights=1234, seed_rays=7, n_voxel=20 ** 3, grid=np.array(cfg.grid), rays=rays.numpy())

    # --- stage vectors -------------------------------------------------------------------------
    S = 24
    xyz, z, _ = model.sample_ray_exp(rays[:, :3], rays[:, 3:6], is_train=False, N_samples=S)
    c7 = coords.from_cartesian(xyz)
    c7n = coords.normalize_coord(c7, downsample=2)
    sf = model.compute_densityfeature(c7n)
    sfc = model.compute_coarse_densityfeature(c7n)
    sigma = model.feature2density(sf)
    dists = torch.cat([z[:, 1:] - z[:, :-1], z[:, -1:] - z[:, -2:-1]], -1)
    alpha, w, bgw = raw2alpha(sigma, dists * model.distance_scale)
    af = model.compute_appfeature(c7n)
    vd = rays[:, 3:6].view(-1, 1, 3).expand(xyz.shape)
    rgb_s = model.renderModule(c7n, vd, af)
    fx.update(st_xyz=np_(xyz), st_z=np_(z), st_c7=np_(c7), st_c7n=np_(c7n), st_sigma_feat=np_(sf),
              st_sigma_feat_coarse=np_(sfc), st_sigma=np_(sigma), st_alpha=np_(alpha), st_weight=np_(w),
              st_bg_weight=np_(bgw), st_app_feat=np_(af), st_rgb_samples=np_(rgb_s))

    # random [M,7] incl. out-of-range coordinates (zero padding) for the lookup ops
    M = 512
    u = torch.from_numpy(synth.hash_uniform(99, 0, M * 7).reshape(M, 7).astype(np.float32))
    q = u * 2.6 - 1.3
    q[:, 6] = (u[:, 6] > 0.5).float()
    fx.update(lk_coords=np_(q), lk_density=np_(model.compute_densityfeature(q)),
              lk_density_coarse=np_(model.compute_coarse_densityfeature(q)), lk_app=np_(model.compute_appfeature(q)))

    # --- end-to-end, eval ------------------------------------------------------------------------
    o = run_forward(model, rays, n_coarse=24, n_fine=0, resampling=False)
    fx.update(e2e_nr_rgb=np_(o[0]), e2e_nr_depth=np_(o[1]), e2e_nr_alpha=np_(o[4]))
    o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)
    fx.update(e2e_rs_rgb=np_(o[0]), e2e_rs_depth=np_(o[1]), e2e_rs_alpha=np_(o[4]))
    o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=False)
    fx.update(e2e_rsf_rgb=np_(o[0]), e2e_rsf_depth=np_(o[1]))

    # --- end-to-end, train noise pinned ------------------------------------------------------------
    jit = torch.from_numpy(synth.hash_uniform(5, 0, 64 * 16).reshape(64, 16).astype(np.float32))
    uu = torch.from_numpy(synth.hash_uniform(5, 1, 64 * 16).reshape(64, 16).astype(np.float32))
    with patched_rand([jit], [uu]):
        o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True, is_train=True)
    fx.update(tr_jitter=np_(jit), tr_u=np_(uu), tr_rgb=np_(o[0]), tr_depth=np_(o[1]))

    # --- backward: MSE grads wrt every parameter (config 4 oracle) -----------------------------------
    gt = torch.from_numpy(synth.hash_uniform(6, 0, 64 * 3).reshape(64, 3).astype(np.float32))
    model.zero_grad()
    with patched_rand([jit], [uu]):
        o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True, is_train=True)
    loss = torch.mean((o[0] - gt) ** 2)
    loss.backward()
    fx.update(bw_gt=np_(gt), bw_loss=np.float32(loss.item()))
    for k, p in model.named_parameters():
        fx["bw_grad/" + k] = np_(p.grad if p.grad is not None else torch.zeros_like(p))
    np.savez_compressed(os.path.join(OUT, "tiny.npz"), **fx)

    # --- envmap variant -----------------------------------------------------------------------------
    cfg_e = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=16)
    w_e = synth.make_weights(cfg_e, seed=1234)
    model_e, _ = build_reference(cfg_e, w_e)
    o = run_forward(model_e, rays, n_coarse=24, n_fine=0, resampling=False)
    rad = model_e.envmap.get_radiance(rays[:, 3:6])
    np.savez_compressed(os.path.join(OUT, "tiny_envmap.npz"), rays=rays.numpy(), seed_weights=1234, envmap_res_H=16,
                        rgb=np_(o[0]), depth=np_(o[1]), bg=np_(o[2]), env=np_(o[3]), alpha=np_(o[4]), radiance=np_(rad))


def capture_stages():
    """Scene-independent known answers: schedules, r normalisation, borders, sample_pdf, samplers."""
    fx = {}
    for name, (near, far, r0) in dict(indoor=(0.01, 15.0, 0.03), ricoh=(0.1, 300.0, 0.05), mid=(0.01, 50.0, 0.05)).items():
        cfg = synth.SceneConfig(n_voxel=20 ** 3, near=near, far=far, r0=r0)
        model, coords = build_reference(cfg, synth.make_weights(cfg, seed=3))
        o = torch.zeros(1, 3)
        d = torch.tensor([[0.0, 0.0, 1.0]])
        for S in (32, 64, 128, 256, 512):
            _, z, _ = model.sample_ray_exp(o, d, is_train=False, N_samples=S)
            fx[f"sched/{name}/{S}"] = np_(z[0])
    # r normalisation sweep on the full-resolution grid (N_r=150, far_r=26.85) and the tiny one
    for name, nv in (("full", 27_000_000), ("tiny", 20 ** 3)):
        cfg = synth.SceneConfig(n_voxel=nv)
        aabb = torch.from_numpy(cfg.aabb)
        with contextlib.redirect_stdout(io.StringIO()):
            coords = YinYangSphericalCoords("cpu", aabb, exp_r=True, N_voxel=nv, r0=cfg.r0, interval_th=True)
        r = torch.cat([torch.linspace(0, 0.2, 257), torch.linspace(0.2, 30.0, 1025),
                       torch.from_numpy((synth.hash_uniform(11, 0, 512) * 27).astype(np.float32))])
        fx[f"normr/{name}/r"] = np_(r)
        fx[f"normr/{name}/out"] = np_(coords.normalize_r(r))
        fx[f"normr/{name}/far_r"] = np_(coords.far[0])
        # from_cartesian + normalize_coord at r=0, poles, region borders and random points
        pts = [[0, 0, 0], [0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [1, 1, 0], [-1, 1e-3, 0],
               [-1, -1e-3, 0], [1, 0, 1], [1, 0, -1], [-1, 1, 0], [-1, -1, 0], [0.3, -0.2, 5], [7, 8, -9], [20, 20, 20]]
        rnd = (synth.hash_uniform(12, 0, 300).reshape(100, 3) * 2 - 1) * 16
        P = torch.cat([torch.tensor(pts, dtype=torch.float32), torch.from_numpy(rnd.astype(np.float32))])
        c7 = coords.from_cartesian(P)
        fx[f"cart/{name}/xyz"] = np_(P)
        fx[f"cart/{name}/c7"] = np_(c7)
        fx[f"cart/{name}/c7n"] = np_(coords.normalize_coord(c7))
    # sample_pdf eval + pinned-u train
    bins = torch.sort(torch.from_numpy(synth.hash_uniform(13, 0, 8 * 31).reshape(8, 31).astype(np.float32)) * 10, -1)[0]
    wts = torch.from_numpy((synth.hash_uniform(13, 1, 8 * 30).reshape(8, 30) ** 4).astype(np.float32))
    wts[3] = 0
    fx["pdf/bins"], fx["pdf/weights"] = np_(bins), np_(wts)
    fx["pdf/eval32"] = np_(sample_pdf(bins, wts, 32, is_train=False))
    uu = torch.from_numpy(synth.hash_uniform(13, 2, 8 * 20).reshape(8, 20).astype(np.float32))
    with patched_rand([], [uu]):
        fx["pdf/train20"] = np_(sample_pdf(bins, wts, 20, is_train=True))
    fx["pdf/u"] = np_(uu)
    # positional encoding
    x = torch.from_numpy((synth.hash_uniform(14, 0, 5 * 4) * 2 - 1).reshape(5, 4).astype(np.float32))
    fx["pe/x"], fx["pe/out"] = np_(x), np_(positional_encoding(x, 2))
    # samplers (SURVEY 8c viii)
    np.random.seed(20221028)
    s = ref_sampler.SimpleSampler(10, 4)
    fx["sampler/simple"] = np.stack([s.nextids().numpy() for _ in range(5)])
    np.random.seed(20221028)
    t = ref_sampler.ThetaImportanceSampler(5, 3, (8, 4), 6, [0, 1, 0, 1])
    fx["sampler/theta_weight"] = t.weight
    fx["sampler/theta_ids"] = np.asarray(t.nextids())
    np.savez_compressed(os.path.join(OUT, "stages.npz"), **fx)


def capture_full():
    """Full barbershop grid [150,172,516]: seeds + outputs only (weights are regenerated by seed)."""
    cfg = synth.SceneConfig()
    weights = synth.make_weights(cfg, seed=1234)
    model, _ = build_reference(cfg, weights)
    rays = torch.from_numpy(synth.make_rays(256, seed=1))
    fx = dict(seed_weights=1234, seed_rays=1, n_rays=256, grid=np.array(cfg.grid))
    o = run_forward(model, rays, n_coarse=64, n_fine=0, resampling=False)
    fx.update(nr64_rgb=np_(o[0]), nr64_depth=np_(o[1]), nr64_alpha=np_(o[4]))
    o = run_forward(model, rays[:64], n_coarse=512, n_fine=0, resampling=False)
    fx.update(nr512_rgb=np_(o[0]), nr512_depth=np_(o[1]))
    o = run_forward(model, rays, n_coarse=32, n_fine=32, resampling=True, use_coarse_sample=True)
    fx.update(rs32_rgb=np_(o[0]), rs32_depth=np_(o[1]))
    o = run_forward(model, rays[:64], n_coarse=128, n_fine=128, resampling=True, use_coarse_sample=True)
    fx.update(rs128_rgb=np_(o[0]), rs128_depth=np_(o[1]))
    np.savez_compressed(os.path.join(OUT, "full.npz"), **fx)


def erp_subset(H: int, W: int):
    """Strided subset of an H x W equirectangular image that keeps the poles, the phi = +-pi seam (columns 0 / W-1), the image
    centre and the yin/yang border latitudes/longitudes: flat ray indices, ~500 rays."""
    rows = sorted(set([0, 1, 2, H // 4 - 1, H // 4, H // 4 + 1, H // 2 - 1, H // 2, 3 * H // 4 - 1, 3 * H // 4, H - 2, H - 1]) | set(range(5, H, 97)))
    cols = sorted(set([0, 1, W // 4 - 1, W // 4, W // 2 - 1, W // 2, 3 * W // 4 - 1, 3 * W // 4, W - 2, W - 1]) | set(range(7, W, 173)))
    rr, cc = np.meshgrid(np.array(rows), np.array(cols), indexing="ij")
    return (rr * W + cc).reshape(-1).astype(np.int64)


def ricoh_poses():
    """Identity at the origin + a general rotation (all nine entries non-trivial) with an off-centre camera position inside the
    0.5 trajectory radius; [2,3,4] float32 camera-to-world matrices."""
    from scipy.spatial.transform import Rotation
    R1 = Rotation.from_rotvec(np.array([0.3, -1.1, 0.5])).as_matrix()
    P = np.zeros((2, 3, 4), np.float32)
    P[0, :, :3] = np.eye(3)
    P[1, :, :3] = R1.astype(np.float32)
    P[1, :, 3] = [0.21, -0.13, 0.30]
    return P


def capture_ricoh():
    """BASELINE config 3 (Ricoh360 scene: configs/EgoNeRF/ricoh/common.txt:5-13 + common.txt): near_far [0.1, 300], r0 0.05,
    density_shift -10, envmap 3 x 3840 x 1920, full [150,172,516] grid; rays from the reference's own ERP generator
    (dataLoader/ray_utils.py:24-40 get_ray_directions_360 + the dataset's normalisation, dataset_egocentric_video.py:57-58, +
    :85-113 get_rays) for 1024 x 2048 images; a ~500-ray subset rendered at 128+128 (default) and 512 samples.
    Seeds + outputs only (weights and the envmap are regenerated by seed)."""
    from dataLoader.ray_utils import get_ray_directions_360, get_rays
    H, W = 1024, 2048
    cfg = synth.SceneConfig(**synth.RICOH)
    weights = synth.make_weights(cfg, seed=1234)
    model, _ = build_reference(cfg, weights)
    dirs = get_ray_directions_360(H, W)
    dirs_n = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    idx = erp_subset(H, W)
    poses = ricoh_poses()
    fx = dict(seed_weights=1234, H=H, W=W, idx=idx, poses=poses, grid=np.array(cfg.grid))
    # a larger slice for the ray generator alone: the subset + two full rows + two full columns
    gen_idx = np.unique(np.concatenate([idx, np.arange(W) + 300 * W, np.arange(W) + (H - 1) * W, np.arange(H) * W, np.arange(H) * W + 1234]))
    fx["gen_idx"] = gen_idx
    for k in range(2):
        c2w = torch.from_numpy(poses[k])
        o_raw, d_raw = get_rays(dirs, c2w)      # un-normalised camera directions
        o, d = get_rays(dirs_n, c2w)            # what the datasets feed the renderer
        fx[f"gen_rays_raw/{k}"] = np_(torch.cat([o_raw, d_raw], 1)[gen_idx])
        fx[f"gen_rays/{k}"] = np_(torch.cat([o, d], 1)[gen_idx])
        rays = torch.cat([o, d], 1)[idx].contiguous()
        fx[f"rays/{k}"] = np_(rays)
        out = run_forward(model, rays, n_coarse=128, n_fine=128, resampling=True, use_coarse_sample=True)
        fx.update({f"rs128/{k}/rgb": np_(out[0]), f"rs128/{k}/depth": np_(out[1]), f"rs128/{k}/bg": np_(out[2]),
                   f"rs128/{k}/env": np_(out[3]), f"rs128/{k}/acc_alpha_sum": np_(out[4].sum(-1))})
        out = run_forward(model, rays, n_coarse=512, n_fine=0, resampling=False)
        fx.update({f"nr512/{k}/rgb": np_(out[0]), f"nr512/{k}/depth": np_(out[1]), f"nr512/{k}/bg": np_(out[2]),
                   f"nr512/{k}/env": np_(out[3])})
    # schedule / LUT known answers of this scene (far_r = 520.5, z up to ~300)
    _, z, _ = model.sample_ray_exp(torch.zeros(1, 3), torch.tensor([[0.0, 0.0, 1.0]]), is_train=False, N_samples=128)
    fx["sched128"] = np_(z[0])
    fx["far_r"] = np_(model.coordinates.far[0])
    np.savez_compressed(os.path.join(OUT, "ricoh.npz"), **fx)


def capture_envmap_full():
    """EnvironmentMap.get_radiance (models/envmap.py:6-34) at the shipped sizes h = 1000 (opt.py default) and h = 1920
    (ricoh/common.txt:10) on a white-noise emission map (every texel independent, so any index slip shows), regenerated by
    seed; directions: axis-aligned ones (poles of the map, the atan2 seam), and hashed random ones."""
    special = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [-1, 1e-7, 0], [-1, -1e-7, 0],
               [-1, 0, 1e-3], [1, 1, 1], [-1, -1, -1], [0.3, -0.2, 0.93], [2.0, 0.0, 0.0], [0, 3.0, 4.0]]
    rnd = synth.hash_uniform(51, 0, 2048 * 3).reshape(2048, 3) * 2 - 1
    d = torch.cat([torch.tensor(special, dtype=torch.float32), torch.from_numpy(rnd.astype(np.float32))])
    fx = dict(dirs=np_(d), seed=52)
    for h in (1000, 1920):
        em = synth.white_envmap(52, h)
        env = EnvironmentMap(h=4, init_strategy="zero", device="cpu")
        env.load_envmap(em, device="cpu")
        fx[f"radiance/{h}"] = np_(env.get_radiance(d))
    np.savez_compressed(os.path.join(OUT, "envmap_full.npz"), **fx)


def capture_alpha_mask():
    """Occupancy semantics: updateAlphaMask + sample_alpha on the tiny grid."""
    import warnings
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    model, coords = build_reference(cfg, synth.make_weights(cfg, seed=1234))
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.updateAlphaMask(tuple(cfg.grid))
    am = model.alphaMask
    M = 400
    u = torch.from_numpy(synth.hash_uniform(21, 0, M * 7).reshape(M, 7).astype(np.float32))
    q = u * 2.4 - 1.2
    q[:, 6] = (u[:, 6] > 0.5).float()
    np.savez_compressed(os.path.join(OUT, "alpha_mask.npz"), seed_weights=1234,
                        step_size=np_(model.stepSize), vol_yin=np_(am.alpha_volume_yin).astype(np.uint8),
                        vol_yang=np_(am.alpha_volume_yang).astype(np.uint8), coords=np_(q), sampled=np_(am.sample_alpha(q)))


def capture_checkpoint():
    """A `.th` checkpoint written by the reference's own EgoNeRF.save (EgoNeRF.py:158-172): tiny grid, envmap and a
    packed alpha mask, so loading exercises every key of the format (state_dict, kwargs with the pickled coordinates /
    envmap objects, alphaMask_{yin,yang}.{shape,mask}, envmap.emission, global_step).  Plus the reference's render of it."""
    import warnings
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=16)
    model, _ = build_reference(cfg, synth.make_weights(cfg, seed=77))
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.updateAlphaMask(tuple(cfg.grid))
    path = os.path.join(OUT, "reference_ckpt.th")
    model.save(path, global_step=4321)
    rays = torch.from_numpy(synth.make_rays(48, seed=17))
    o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)
    np.savez_compressed(os.path.join(OUT, "reference_ckpt_render.npz"), rays=rays.numpy(), rgb=np_(o[0]), depth=np_(o[1]),
                        bg=np_(o[2]), env=np_(o[3]), alpha=np_(o[4]))


def capture_train_extras():
    """Training-step pieces beyond the MSE gradient (train.py:245-330): ray-entropy gradient through `alpha` with the envmap's
    ones column, envmap-emission gradient, envmap pre-training (train.py:218-236), the TV / L1 / ortho regularisers
    (EgoNeRF.py:191-230, utils.py:155-171) with their gradients, and upsample_volume_grid (EgoNeRF.py:415-435)."""
    from utils import TVLoss, ray_entropy_loss
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=16)
    w = synth.make_weights(cfg, seed=4321)
    model, coords = build_reference(cfg, w)
    rays = torch.from_numpy(synth.make_rays(64, seed=11))
    jit = torch.from_numpy(synth.hash_uniform(8, 0, 64 * 16).reshape(64, 16).astype(np.float32))
    uu = torch.from_numpy(synth.hash_uniform(8, 1, 64 * 16).reshape(64, 16).astype(np.float32))
    gt = torch.from_numpy(synth.hash_uniform(8, 2, 64 * 3).reshape(64, 3).astype(np.float32))
    fx = dict(seed_weights=4321, seed_rays=11, envmap_res_H=16, rays=rays.numpy(), jitter=np_(jit), u=np_(uu), gt=np_(gt),
              entropy_weight=np.float32(0.05))

    def grads(prefix):
        for k, p in model.named_parameters():
            fx[f"{prefix}/{k}"] = np_(p.grad if p.grad is not None else torch.zeros_like(p))
        em = model.envmap.emission
        fx[f"{prefix}/envmap.emission"] = np_(em.grad if em.grad is not None else torch.zeros_like(em))

    def zero():
        model.zero_grad()
        model.envmap.emission.grad = None

    # (1) MSE + entropy, train noise pinned, envmap on
    zero()
    with patched_rand([jit], [uu]):
        o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True, is_train=True)
    mse = torch.mean((o[0] - gt) ** 2)
    ent = ray_entropy_loss(o[4])
    (mse + 0.05 * ent).backward()
    fx.update(ent_rgb=np_(o[0]), ent_alpha=np_(o[4]), ent_mse=np.float32(mse.item()), ent_entropy=np.float32(ent.item()))
    grads("ent_grad")
    # entropy alone (isolates the alpha path)
    zero()
    with patched_rand([jit], [uu]):
        o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True, is_train=True)
    ray_entropy_loss(o[4]).backward()
    grads("entonly_grad")

    # (2) envmap pre-training
    zero()
    env = model(rays_chunk=rays, pretrain_envmap=True)
    lp = torch.mean((env - gt) ** 2)
    lp.backward()
    fx.update(pre_env=np_(env), pre_loss=np.float32(lp.item()), pre_grad=np_(model.envmap.emission.grad))

    # (3) regularisers
    tv = TVLoss()
    for name, fn in (("tv_density", lambda: model.TV_loss_density(tv)), ("tv_app", lambda: model.TV_loss_app(tv)),
                     ("l1", model.density_L1), ("ortho", model.vector_comp_diffs)):
        zero()
        v = fn()
        v.backward()
        fx[f"reg/{name}/value"] = np.float32(v.item())
        for k, p in model.named_parameters():
            if p.grad is not None:
                fx[f"reg/{name}/grad/{k}"] = np_(p.grad)

    # (4) coarse-to-fine upsampling, then a render on the finer grid
    target = [20, 22, 64]
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        model.upsample_volume_grid(list(target))
        coords.set_resolution(list(target))
        model.update_coarse_sigma_grid()
    fx["up_target"] = np.array(target)
    for k, p in model.named_parameters():
        if "plane" in k or "line" in k:
            fx[f"up/{k}"] = np_(p)
    o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)
    fx.update(up_rgb=np_(o[0]), up_depth=np_(o[1]))
    np.savez_compressed(os.path.join(OUT, "train_extras.npz"), **fx)


def capture_metrics():
    """utils.py:104-152 rgb_ssim as renderer.py:160 calls it (torch float32 images, max_val 1) + the PSNR line (:156-157)."""
    from utils import rgb_ssim
    H, W = 40, 56
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([0.5 + 0.4 * np.sin(xx / 5.0 + c) * np.cos(yy / 7.0 - c) for c in range(3)], -1)
    noise = synth.hash_uniform(31, 0, H * W * 3).reshape(H, W, 3)
    img0 = torch.from_numpy(np.clip(base, 0, 1).astype(np.float32))
    img1 = torch.from_numpy(np.clip(base + 0.15 * (noise - 0.5), 0, 1).astype(np.float32))
    flat = torch.full((H, W, 3), 0.25)
    fx = dict(img0=img0.numpy(), img1=img1.numpy())
    fx["ssim"] = np.float64(rgb_ssim(img0, img1, 1))
    fx["ssim_map"] = np.asarray(rgb_ssim(img0, img1, 1, return_map=True))
    fx["ssim_same"] = np.float64(rgb_ssim(img0, img0, 1))
    fx["ssim_flat"] = np.float64(rgb_ssim(flat, img1, 1))
    fx["ssim_fs7"] = np.float64(rgb_ssim(img0, img1, 1, filter_size=7, filter_sigma=1.0))
    loss = torch.mean((img1 - img0) ** 2)
    fx["psnr"] = np.float64(-10.0 * np.log(loss.item()) / np.log(10.0))
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **fx)


def capture_ws():
    """extra/ws_ssim.py:12-33: the cos-latitude row weights (generate_ws / estws) and the weighted mean of an SSIM map exactly as
    ws_ssim forms it (`np.sum(map * ws) / ws.sum()`).  torchmetrics is absent, so the map fed to that expression is the
    reference's own utils.rgb_ssim(..., return_map=True) map (channel mean), not torchmetrics' padded one."""
    import importlib
    _stub("torchmetrics", StructuralSimilarityIndexMeasure=None)
    if "scipy.constants.constants" not in sys.modules:
        try:
            importlib.import_module("scipy.constants.constants")
        except Exception:
            _stub("scipy.constants.constants", pi=np.pi)
    ws_mod = importlib.import_module("extra.ws_ssim")
    from utils import rgb_ssim
    mx = np.load(os.path.join(OUT, "metrics.npz"))
    img0, img1 = torch.from_numpy(mx["img0"]), torch.from_numpy(mx["img1"])
    smap = np.asarray(rgb_ssim(img0, img1, 1, return_map=True)).mean(-1)     # [Ho, Wo]
    ws = ws_mod.estws(smap)
    fx = dict(ws_30x46=ws, wsssim=np.float64(np.sum(smap * ws) / ws.sum()), ssim_map_mean=smap)
    for n in (7, 64, 1024):
        fx[f"ws_rows/{n}"] = ws_mod.estws(np.zeros((n, 3)))[:, 0]
    np.savez_compressed(os.path.join(OUT, "ws_metrics.npz"), **fx)


def capture_skip():
    """Skip semantics of TensorBase.forward (models/tensorBase.py:438-510), which is the only place the reference applies an
    alpha mask and the rayMarch_weight_thres appearance skip — run through TensorVMSplit (models/tensoRF.py:127-284) with
    CartesianCoords, an AlphaGridMask over a hashed {0,1} volume and a large threshold so that both branches bite.
    Stored: the dense per-sample quantities the skip logic consumes (density feature, mask lookups, inside-aabb flags,
    distances, per-sample colours) and what forward returns, for exp_sampling False/True.  The oracle's restatement of the
    skip logic is checked against these; the HIP path is then compared with that restatement on the EgoNeRF field."""
    from models.tensoRF import TensorVMSplit
    from models.tensorBase import AlphaGridMask
    from models.coordinates import CartesianCoords
    torch.manual_seed(7)
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    grid = [16, 16, 16]
    with contextlib.redirect_stdout(io.StringIO()):
        coords = CartesianCoords("cpu", aabb)
        coords.invgridSize = 1.0 / (aabb[1] - aabb[0])  # what train.py's dataset plumbing leaves on the object (coordinates.py:50)
        model = TensorVMSplit(aabb, grid, "cpu", coords, density_n_comp=[4, 4, 4], appearance_n_comp=[8, 8, 8], app_dim=27,
                              near_far=[0.05, 2.5], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=-1.0,
                              distance_scale=25, rayMarch_weight_thres=2e-2, pos_pe=6, view_pe=2, fea_pe=2, featureC=32,
                              step_ratio=0.5, fea2denseAct="softplus")
    with torch.no_grad():
        for p in list(model.density_plane) + list(model.density_line):
            p.mul_(8.0)  # enough density for rays to saturate
        for p in list(model.app_plane) + list(model.app_line) + [model.basis_mat.weight] + [m.weight for m in model.renderModule.mlp if hasattr(m, "weight")]:
            p.mul_(3.0)  # a colour range worth comparing
    # blocky occupancy: a hashed 4^3 lattice blown up to 16^3 (a per-voxel random volume is "occupied" almost everywhere once
    # it is looked up trilinearly)
    coarse = (synth.hash_uniform(61, 0, 4 ** 3) > 0.45).astype(np.float32).reshape(4, 4, 4)
    vol = torch.from_numpy(np.kron(coarse, np.ones((4, 4, 4), np.float32)))
    model.alphaMask = AlphaGridMask("cpu", vol)
    model.eval()
    rays = torch.from_numpy(synth.make_rays(48, seed=23, origin_extent=0.4))
    fx = dict(rays=rays.numpy(), weight_thres=np.float32(2e-2), distance_scale=np.float32(25), density_shift=np.float32(-1.0),
              mask_volume=vol.numpy().astype(np.uint8))
    for tag, exp in (("uni", False), ("exp", True)):
        S = 40
        with torch.no_grad():
            sampler = model.sample_ray_exp if exp else model.sample_ray
            xyz, z, ray_valid = sampler(rays[:, :3], rays[:, 3:6], is_train=False, N_samples=S)
            if z.shape[0] == 1:
                z = z.expand(rays.shape[0], S)
            c = coords.normalize_coord(coords.from_cartesian(xyz))
            flat = c.reshape(-1, 3)
            mask_alpha = model.alphaMask.sample_alpha(flat).view(c.shape[:2])
            sigma_feat = model.compute_densityfeature(flat).view(c.shape[:2])
            vd = rays[:, 3:6].view(-1, 1, 3).expand(xyz.shape)
            rgb_dense = model.renderModule(flat, vd.reshape(-1, 3), model.compute_appfeature(flat)).view(*c.shape[:2], 3)
            out = model(rays, is_train=False, white_bg=False, ndc_ray=False, N_samples=S, exp_sampling=exp)
            model.alphaMask, keep = None, model.alphaMask
            out_nomask = model(rays, is_train=False, white_bg=False, ndc_ray=False, N_samples=S, exp_sampling=exp)
            model.alphaMask = keep
        fx.update({f"{tag}/z": np_(z), f"{tag}/ray_valid": np_(ray_valid), f"{tag}/mask_alpha": np_(mask_alpha),
                   f"{tag}/sigma_feat": np_(sigma_feat), f"{tag}/rgb_dense": np_(rgb_dense),
                   f"{tag}/rgb": np_(out[0]), f"{tag}/depth": np_(out[1]), f"{tag}/alpha": np_(out[4]),
                   f"{tag}/nomask_rgb": np_(out_nomask[0]), f"{tag}/nomask_depth": np_(out_nomask[1]), f"{tag}/nomask_alpha": np_(out_nomask[4])})
    np.savez_compressed(os.path.join(OUT, "skip_semantics.npz"), **fx)


def capture_omniblender():
    """dataLoader/dataset_omniblender.py:11-95 on a three-frame synthetic dataset (transform.json + split lists + 8 x 4 PNGs written
    to a temp dir; downsample 250 -> img_wh (8, 4)): poses, centre, scene_bbox, radius, all_rays (both is_stack settings), all_rgbs.
    The fixture keeps the frame matrices / file names / pixels so the test can rebuild the same directory."""
    import json, tempfile
    from PIL import Image
    _tv.transforms.ToTensor = lambda: (lambda img: torch.from_numpy(np.asarray(img)).permute(2, 0, 1).float().div(255.0))
    from dataLoader.dataset_omniblender import OmniBlenderDataset
    from scipy.spatial.transform import Rotation
    frames, pix = [], {}
    for k in range(3):
        M = np.eye(4)
        M[:3, :3] = Rotation.from_rotvec(np.array([0.2 * k, 0.5 - 0.3 * k, 0.1 + 0.4 * k])).as_matrix()
        M[:3, 3] = [0.3 * np.cos(2.0 * k), 0.05 * k, 0.3 * np.sin(2.0 * k)]
        frames.append(dict(file_path=f"{k:04d}.png", transform_matrix=M.tolist()))
        pix[f"{k:04d}"] = (synth.hash_uniform(71 + k, 0, 4 * 8 * 4).reshape(4, 8, 4) * 255).astype(np.uint8)   # RGBA
    fx = dict(frames_json=np.array(json.dumps(dict(indoor=True, frames=frames))), train_list=np.array("0000\n0002\n"), test_list=np.array("0001\n"))
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "images"))
        json.dump(dict(indoor=True, frames=frames), open(os.path.join(d, "transform.json"), "w"))
        open(os.path.join(d, "train.txt"), "w").write("0000\n0002\n")
        open(os.path.join(d, "test.txt"), "w").write("0001\n")
        for name, a in pix.items():
            Image.fromarray(a, "RGBA").save(os.path.join(d, "images", name + ".png"))
            fx["png/" + name] = a
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            for split, stack in (("train", False), ("test", True)):
                ds = OmniBlenderDataset(data_dir=d, split=split, near_far=[0.01, 15.0], downsample=250.0, is_stack=stack)
                fx.update({f"{split}/poses": np_(ds.poses), f"{split}/center": np_(ds.center), f"{split}/scene_bbox": np_(ds.scene_bbox),
                           f"{split}/radius": np_(ds.radius), f"{split}/all_rays": np_(ds.all_rays), f"{split}/all_rgbs": np_(ds.all_rgbs),
                           f"{split}/img_wh": np.array(ds.img_wh)})
            ds = OmniBlenderDataset(data_dir=d, split="train", near_far=[0.01, 15.0], downsample=250.0, roi=[0.25, 1.0, 0.0, 0.5])
            fx["roi/all_rays"] = np_(ds.all_rays)
    np.savez_compressed(os.path.join(OUT, "omniblender.npz"), **fx)


def capture_plainexp():
    """interval_th=False: the plain exponential r grid (coordinates.py:132-155, with the `downsample=2` the forward passes,
    EgoNeRF.py:524) and the matching sample schedule (EgoNeRF.py:59-67).  Eval mode."""
    cfg = synth.SceneConfig(n_voxel=20 ** 3, interval_th=False)
    w = synth.make_weights(cfg, seed=1234)
    model, coords = build_reference(cfg, w)
    rays = torch.from_numpy(synth.make_rays(64, seed=7))
    fx = dict(seed_weights=1234, seed_rays=7, rays=rays.numpy())
    for S in (16, 24, 64):
        _, z, _ = model.sample_ray_exp(rays[:1, :3], rays[:1, 3:6], is_train=False, N_samples=S)
        fx[f"sched/{S}"] = np_(z[0])
    r = torch.cat([torch.linspace(1e-4, 30.0, 4001), torch.tensor([0.03, 0.0300001, 26.8, 26.9])])
    fx["normr/r"] = np_(r)
    fx["normr/out"] = np_(coords.normalize_r(r))
    fx["normr/out_ds2"] = np_(coords.normalize_r(r, downsample=2))
    xyz, z, _ = model.sample_ray_exp(rays[:, :3], rays[:, 3:6], is_train=False, N_samples=24)
    c7 = coords.from_cartesian(xyz)
    fx["c7n_ds2"] = np_(coords.normalize_coord(c7, downsample=2))
    fx["c7n"] = np_(coords.normalize_coord(c7))
    o = run_forward(model, rays, n_coarse=24, n_fine=0, resampling=False)
    fx.update(nr_rgb=np_(o[0]), nr_depth=np_(o[1]), nr_alpha=np_(o[4]))
    o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)
    fx.update(rs_rgb=np_(o[0]), rs_depth=np_(o[1]))
    # is_train: the noise is added to the exponent (EgoNeRF.py:64-67) and the distances are an exclusive prefix sum by matmul
    jit = torch.from_numpy(synth.hash_uniform(13, 0, 64 * 16).reshape(64, 16).astype(np.float32))
    uu = torch.from_numpy(synth.hash_uniform(13, 1, 64 * 16).reshape(64, 16).astype(np.float32))
    with patched_rand([jit.clone()], []):
        _, z, _ = model.sample_ray_exp(rays[:, :3], rays[:, 3:6], is_train=True, N_samples=16)
    with patched_rand([jit.clone()], [uu]):
        o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True, is_train=True)
    fx.update(tr_jitter=np_(jit), tr_u=np_(uu), tr_z=np_(z), tr_rgb=np_(o[0]), tr_depth=np_(o[1]))
    np.savez_compressed(os.path.join(OUT, "tiny_plainexp.npz"), **fx)
    # coarse-to-fine upsampling on the plain exponential grid (coordinates.py:260-262), then a render on the finer grid
    target = [20, 22, 64]
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        model.upsample_volume_grid(list(target))
        coords.set_resolution(list(target))
        model.update_coarse_sigma_grid()
    up = dict(seed_weights=1234, seed_rays=7, rays=rays.numpy(), up_target=np.array(target))
    for k, p in model.named_parameters():   # density tables + every line; the (large) appearance planes are pinned through the render below
        if ("plane" in k and "density" in k) or "line" in k:
            up[f"up/{k}"] = np_(p)
    o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)
    up.update(up_rgb=np_(o[0]), up_depth=np_(o[1]))
    np.savez_compressed(os.path.join(OUT, "tiny_plainexp_up.npz"), **up)


def capture_sh():
    """models/tensorBase.py:30-34 SHRender on flat inputs (the only way the reference's function can be called)."""
    from models.tensorBase import SHRender
    d = torch.nn.functional.normalize(torch.from_numpy(synth.hash_uniform(41, 0, 300 * 3).reshape(300, 3).astype(np.float32) * 2 - 1), dim=-1)
    f = torch.from_numpy(synth.hash_uniform(41, 1, 300 * 27).reshape(300, 27).astype(np.float32) * 4 - 2)
    np.savez_compressed(os.path.join(OUT, "sh_render.npz"), dirs=d.numpy(), features=f.numpy(), rgb=np_(SHRender(None, d, f)))


def capture_uniform():
    """exp_sampling=False: TensorBase.sample_ray (tensorBase.py:308-327) feeding EgoNeRF.forward — per-ray aabb entry + uniform
    steps of stepSize; eval (both resampling modes) and train with the noise pinned."""
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    w = synth.make_weights(cfg, seed=1234)
    model, coords = build_reference(cfg, w)
    rays = torch.from_numpy(synth.make_rays(64, seed=7))
    fx = dict(seed_weights=1234, seed_rays=7, rays=rays.numpy(), step_size=np.float32(float(model.stepSize)))

    def fwd(**kw):
        with contextlib.redirect_stdout(io.StringIO()):
            return volume_renderer(rays, model, chunk=4096, exp_sampling=False, device="cpu", interval_th=True, **kw)

    xyz, z, _ = model.sample_ray(rays[:, :3], rays[:, 3:6], is_train=False, N_samples=24)
    fx.update(z_eval=np_(z), xyz_eval=np_(xyz))
    o = fwd(n_coarse=24, n_fine=0, resampling=False)
    fx.update(nr_rgb=np_(o[0]), nr_depth=np_(o[1]), nr_alpha=np_(o[4]))
    o = fwd(n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)
    fx.update(rs_rgb=np_(o[0]), rs_depth=np_(o[1]))
    jit = torch.from_numpy(synth.hash_uniform(12, 0, 64 * 16).reshape(64, 16).astype(np.float32))
    uu = torch.from_numpy(synth.hash_uniform(12, 1, 64 * 16).reshape(64, 16).astype(np.float32))
    with patched_rand([jit], [uu]):
        o = fwd(n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True, is_train=True)
    fx.update(tr_jitter=np_(jit), tr_u=np_(uu), tr_rgb=np_(o[0]), tr_depth=np_(o[1]))
    np.savez_compressed(os.path.join(OUT, "tiny_uniform.npz"), **fx)
    # rays that start OUTSIDE the aabb enter it at different distances t_min (clamped to [near, far]): in eval the reference then places
    # the first-pass samples per ray but measures every ray with ray 0's distances (EgoNeRF.py:515-516)
    far = synth.make_rays(64, seed=9)
    far[1::2, :3] = -far[1::2, 3:6] * (19.0 + 6.0 * synth.hash_uniform(14, 0, 32).reshape(32, 1).astype(np.float32)) + 0.5 * far[1::2, :3]
    rays = torch.from_numpy(far)
    xyz, z, _ = model.sample_ray(rays[:, :3], rays[:, 3:6], is_train=False, N_samples=24)
    assert len(torch.unique(z[:, 0])) > 20
    fm = dict(seed_weights=1234, rays=rays.numpy(), z_eval=np_(z))
    o = fwd(n_coarse=24, n_fine=0, resampling=False)
    fm.update(nr_rgb=np_(o[0]), nr_depth=np_(o[1]), nr_alpha=np_(o[4]))
    o = fwd(n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)
    fm.update(rs_rgb=np_(o[0]), rs_depth=np_(o[1]), rs_alpha=np_(o[4]))
    np.savez_compressed(os.path.join(OUT, "tiny_uniform_mixed.npz"), **fm)


SHAPES = {   # model shapes opt.py:87-100 can produce besides the one every shipped config resolves to (16 / 48 / 27 / 128 / 2 / 2)
    "small_head": dict(density_n_comp=(8, 8, 8), app_n_comp=(24, 24, 24), app_dim=27, featureC=64, view_pe=2, fea_pe=2),
    "ctor_defaults": dict(density_n_comp=(16, 16, 16), app_n_comp=(48, 48, 48), app_dim=12, featureC=128, view_pe=6, fea_pe=6),
    "no_encoding": dict(density_n_comp=(24, 24, 24), app_n_comp=(8, 8, 8), app_dim=27, featureC=128, view_pe=0, fea_pe=0),
    "tuned_head_other_density": dict(density_n_comp=(8, 8, 8), app_n_comp=(48, 48, 48), app_dim=27, featureC=128, view_pe=2, fea_pe=2),
}


def capture_shapes():
    """The reference on a tiny grid with head / table shapes other than the shipped one (TensorBase ctor defaults: view_pe = fea_pe =
    6, tensorBase.py:133-139; narrower tables and MLP; no positional encoding): stage ops on random coordinates and the end-to-end
    render in both resampling modes, envmap on for one of them."""
    fx = {}
    for name, kw in SHAPES.items():
        cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=(name == "small_head"), envmap_res_H=16, **kw)
        weights = synth.make_weights(cfg, seed=4321)
        model, coords = build_reference(cfg, weights)
        rays = torch.from_numpy(synth.make_rays(48, seed=17))
        M = 256
        u = torch.from_numpy(synth.hash_uniform(98, 0, M * 7).reshape(M, 7).astype(np.float32))
        q = u * 2.6 - 1.3
        q[:, 6] = (u[:, 6] > 0.5).float()
        af = model.compute_appfeature(q)
        dirs = torch.nn.functional.normalize(torch.from_numpy(synth.hash_uniform(97, 0, M * 3).reshape(M, 3).astype(np.float32)) * 2 - 1, dim=-1)
        fx.update({f"{name}/coords": np_(q), f"{name}/dirs": np_(dirs), f"{name}/density": np_(model.compute_densityfeature(q)),
                   f"{name}/density_coarse": np_(model.compute_coarse_densityfeature(q)), f"{name}/app": np_(af),
                   f"{name}/rgb_samples": np_(model.renderModule(q, dirs, af))})
        o = run_forward(model, rays, n_coarse=24, n_fine=0, resampling=False)
        fx.update({f"{name}/nr_rgb": np_(o[0]), f"{name}/nr_depth": np_(o[1]), f"{name}/nr_alpha": np_(o[4])})
        o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)
        fx.update({f"{name}/rs_rgb": np_(o[0]), f"{name}/rs_depth": np_(o[1])})
    fx["seed_weights"], fx["seed_rays"] = 4321, 17
    np.savez_compressed(os.path.join(OUT, "shapes.npz"), **fx)


def capture_shapes_grad():
    """Autograd of the reference through EgoNeRF.forward (is_train, 16 + 16 resampling, noise pinned; MSE against random targets) for
    model shapes other than the shipped one: gradients of every parameter (train.py:312-314 on opt.py:87-100's other shapes)."""
    fx = {}
    for name in ("small_head", "ctor_defaults", "tuned_head_other_density"):
        cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=(name == "small_head"), envmap_res_H=16, **SHAPES[name])
        weights = synth.make_weights(cfg, seed=4321)
        model, coords = build_reference(cfg, weights)
        model.train()
        rays = torch.from_numpy(synth.make_rays(48, seed=17))
        jit = torch.from_numpy(synth.hash_uniform(15, 0, 48 * 16).reshape(48, 16).astype(np.float32))
        uu = torch.from_numpy(synth.hash_uniform(15, 1, 48 * 16).reshape(48, 16).astype(np.float32))
        gt = torch.from_numpy(synth.hash_uniform(16, 0, 48 * 3).reshape(48, 3).astype(np.float32))
        model.zero_grad()
        with patched_rand([jit], [uu]):
            o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True, is_train=True)
        loss = torch.mean((o[0] - gt) ** 2)
        loss.backward()
        fx.update({f"{name}/jitter": np_(jit), f"{name}/u": np_(uu), f"{name}/gt": np_(gt), f"{name}/rgb": np_(o[0]),
                   f"{name}/loss": np.float32(loss.item())})
        for k, p in model.named_parameters():
            fx[f"{name}/grad/{k}"] = np_(p.grad if p.grad is not None else torch.zeros_like(p))
        if cfg.use_envmap:
            fx[f"{name}/grad/envmap.emission"] = np_(model.envmap.emission.grad)
    fx["seed_weights"], fx["seed_rays"] = 4321, 17
    np.savez_compressed(os.path.join(OUT, "shapes_grad.npz"), **fx)


HEADS = {   # the other appearance heads EgoNeRF.forward runs with (tensorBase.py:186-200); `fea_pe` is what opt.py passes, MLPRender ignores it
    "mlp_head": dict(shadingMode="MLP", app_dim=27, view_pe=2, fea_pe=2, featureC=128),
    "mlp_head_small": dict(shadingMode="MLP", density_n_comp=(8, 8, 8), app_n_comp=(24, 24, 24), app_dim=12, view_pe=6, fea_pe=6, featureC=64),
    "rgb_head": dict(shadingMode="RGB", app_dim=3),
}


def capture_heads():
    """shadingMode 'MLP' (MLPRender, tensorBase.py:107-129) and 'RGB' (RGBRender, :37-39) through the reference's EgoNeRF.forward on the
    tiny grid: per-sample colours, the end-to-end render in both resampling modes (envmap on for the RGB head), and the autograd
    gradients of every parameter for the is_train render with pinned noise + MSE (train.py:312-314)."""
    fx = {}
    for name, kw in HEADS.items():
        cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=(name == "rgb_head"), envmap_res_H=16, **kw)
        weights = synth.make_weights(cfg, seed=2468)
        model, coords = build_reference(cfg, weights)
        rays = torch.from_numpy(synth.make_rays(48, seed=19))
        M = 256
        u = torch.from_numpy(synth.hash_uniform(96, 0, M * 7).reshape(M, 7).astype(np.float32))
        q = u * 2.6 - 1.3
        q[:, 6] = (u[:, 6] > 0.5).float()
        af = model.compute_appfeature(q)
        dirs = torch.nn.functional.normalize(torch.from_numpy(synth.hash_uniform(95, 0, M * 3).reshape(M, 3).astype(np.float32)) * 2 - 1, dim=-1)
        fx.update({f"{name}/coords": np_(q), f"{name}/dirs": np_(dirs), f"{name}/app": np_(af),
                   f"{name}/rgb_samples": np_(model.renderModule(q, dirs, af))})
        o = run_forward(model, rays, n_coarse=24, n_fine=0, resampling=False)
        fx.update({f"{name}/nr_rgb": np_(o[0]), f"{name}/nr_depth": np_(o[1]), f"{name}/nr_alpha": np_(o[4])})
        o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True)
        fx.update({f"{name}/rs_rgb": np_(o[0]), f"{name}/rs_depth": np_(o[1])})
        # autograd
        model.train()
        jit = torch.from_numpy(synth.hash_uniform(25, 0, 48 * 16).reshape(48, 16).astype(np.float32))
        uu = torch.from_numpy(synth.hash_uniform(25, 1, 48 * 16).reshape(48, 16).astype(np.float32))
        gt = torch.from_numpy(synth.hash_uniform(26, 0, 48 * 3).reshape(48, 3).astype(np.float32))
        model.zero_grad()
        with patched_rand([jit], [uu]):
            o = run_forward(model, rays, n_coarse=16, n_fine=16, resampling=True, use_coarse_sample=True, is_train=True)
        loss = torch.mean((o[0] - gt) ** 2)
        loss.backward()
        fx.update({f"{name}/jitter": np_(jit), f"{name}/u": np_(uu), f"{name}/gt": np_(gt), f"{name}/train_rgb": np_(o[0]),
                   f"{name}/loss": np.float32(loss.item())})
        for k, p in model.named_parameters():
            fx[f"{name}/grad/{k}"] = np_(p.grad if p.grad is not None else torch.zeros_like(p))
        if cfg.use_envmap:
            fx[f"{name}/grad/envmap.emission"] = np_(model.envmap.emission.grad)
    fx["seed_weights"], fx["seed_rays"] = 2468, 19
    np.savez_compressed(os.path.join(OUT, "heads.npz"), **fx)


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["tiny", "stages", "full", "alpha_mask", "checkpoint", "train_extras", "metrics", "plainexp", "sh", "uniform", "ricoh", "envmap_full", "ws", "skip", "omniblender", "shapes", "shapes_grad", "heads"]
    for name in which:
        globals()["capture_" + name]()
        print("captured", name)
    assert not os.path.exists(os.path.join(REF, "models", "__pycache__")), "bytecode leaked into the reference mount"
