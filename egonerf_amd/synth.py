"""Deterministic synthetic scenes for parity tests and the bench.

Everything here is integer hashing plus IEEE add/mul (no libm, no BLAS, no torch RNG stream), so the
build container, the GPU box and the golden-capture script all regenerate bit-identical weights and
rays from a seed.  The scene statistics follow SURVEY.md section 8(d): low-frequency random fields so
that rays actually terminate (default 0.1*randn init gives acc ~0.13, useless as a test).

Shapes/keys follow the reference state_dict (models/EgoNeRF.py:102-122, models/tensorBase.py:54-66):
  {density,app}_{plane,line}_{yin,yang}.{0,1,2}, basis_mat_{yin,yang}.weight, renderModule.mlp.{0,2,4}.{weight,bias}
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def hash_uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n uniforms in [0,1) with 24-bit granularity (exact in fp32), float64 array."""
    idx = np.arange(n, dtype=np.uint64)
    key = np.uint64((seed * 0x9E3779B97F4A7C15 + stream * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        x = _mix64(idx * np.uint64(0x2545F4914F6CDD1D) + key)
    return (x >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def hash_normal(seed: int, stream: int, n: int) -> np.ndarray:
    """Approximate N(0,1): Irwin-Hall sum of 4 uniforms, rescaled (variance 4/12 -> 1). No libm."""
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += hash_uniform(seed, stream * 4 + k + 1000003, n)
    return (s - 2.0) * 1.7320508075688772  # sqrt(3)


def _lerp_axis(a: np.ndarray, n_out: int, axis: int) -> np.ndarray:
    """Linear resize along one axis with align_corners=True semantics, explicit gathers only."""
    n_in = a.shape[axis]
    if n_in == n_out:
        return a
    if n_out == 1:
        return np.take(a, [0], axis=axis)
    pos = np.arange(n_out, dtype=np.float64) * ((n_in - 1) / (n_out - 1))
    i0 = np.minimum(np.floor(pos).astype(np.int64), n_in - 2) if n_in > 1 else np.zeros(n_out, np.int64)
    f = pos - i0
    lo = np.take(a, i0, axis=axis)
    hi = np.take(a, np.minimum(i0 + 1, n_in - 1), axis=axis)
    shape = [1] * a.ndim
    shape[axis] = n_out
    f = f.reshape(shape)
    return lo * (1.0 - f) + hi * f


def smooth_field(seed: int, stream: int, C: int, H: int, W: int, lattice: int = 12) -> np.ndarray:
    """(C,H,W) float64: min(lattice,H) x min(lattice,W) N(0,1) lattice, bilinearly up-sampled."""
    lh, lw = min(lattice, H), min(lattice, W)
    base = hash_normal(seed, stream, C * lh * lw).reshape(C, lh, lw)
    return _lerp_axis(_lerp_axis(base, H, 1), W, 2)


MAT_MODE = ((0, 1), (0, 2), (1, 2))  # models/EgoNeRF.py:30
VEC_MODE = (2, 1, 0)  # models/EgoNeRF.py:31


def n_to_reso(n_voxels: float) -> List[int]:
    """Yin-yang grid resolution rule (models/coordinates.py:507-520); keeps Python's inexact pow(x,1/3)."""
    n_r = int(pow(n_voxels, 1 / 3) / 2)
    n_t = int(n_r * 2 * math.sqrt(3) / 3)
    n_p = n_t * 3
    up = lambda v: v + 1 if v % 2 else v
    return [up(n_r), up(n_t), up(n_p)]


@dataclass
class SceneConfig:
    """Resolved scene scalars; defaults = OmniBlender indoor / barbershop."""
    n_voxel: float = 27_000_000
    near: float = 0.01
    far: float = 15.0
    r0: float = 0.03
    traj_radius: float = 0.5
    density_shift: float = -8.0
    distance_scale: float = 25.0
    density_n_comp: Tuple[int, int, int] = (16, 16, 16)
    app_n_comp: Tuple[int, int, int] = (48, 48, 48)
    app_dim: int = 27
    view_pe: int = 2
    fea_pe: int = 2
    featureC: int = 128
    use_envmap: bool = False
    envmap_res_H: int = 1000
    interval_th: bool = True  # configs/EgoNeRF/common.txt:19; False = the plain exponential r grid / sample schedule
    # appearance head (tensorBase.py:186-200): "MLP_Fea" (every shipped config), "MLP" (MLPRender: no feature encoding, `fea_pe` is
    # ignored) or "RGB" (RGBRender: colour = the 3 appearance features, no MLP; needs app_dim == 3)
    shadingMode: str = "MLP_Fea"
    grid: List[int] = field(default_factory=list)

    def __post_init__(self):
        if not self.grid:
            self.grid = n_to_reso(self.n_voxel)

    @property
    def aabb(self) -> np.ndarray:
        e = self.traj_radius + self.far  # dataLoader/dataset_omniblender.py:24-32
        return np.array([[-e, -e, -e], [e, e, e]], dtype=np.float32)

    @property
    def head_fea_pe(self) -> int:
        """Feature-encoding frequencies the head really uses: MLPRender (tensorBase.py:107-129) has none whatever `fea_pe` says."""
        return self.fea_pe if self.shadingMode == "MLP_Fea" else 0

    @property
    def in_mlpC(self) -> int:
        if self.shadingMode == "RGB":
            return 0
        return 2 * self.view_pe * 3 + 2 * self.head_fea_pe * self.app_dim + 3 + self.app_dim


RICOH = dict(near=0.1, far=300.0, r0=0.05, density_shift=-10.0, use_envmap=True, envmap_res_H=1920)


def table_shapes(cfg: SceneConfig, n_comp: Sequence[int]) -> Tuple[list, list]:
    planes, lines = [], []
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        planes.append((1, n_comp[i], cfg.grid[m1], cfg.grid[m0]))
        lines.append((1, n_comp[i], cfg.grid[VEC_MODE[i]], 1))
    return planes, lines


def make_weights(cfg: SceneConfig, seed: int = 1234, mlp_gain: float = 3.0) -> Dict[str, np.ndarray]:
    """Reference-layout fp32 state dict with smooth-field tables and uniform-init linears."""
    out: Dict[str, np.ndarray] = {}
    stream = 0
    for kind, n_comp, scale, white in (("density", cfg.density_n_comp, 0.9, 0.0), ("app", cfg.app_n_comp, 0.5, 0.1)):
        planes, lines = table_shapes(cfg, n_comp)
        for g in ("yin", "yang"):
            for i in range(3):
                for what, shp in (("plane", planes[i]), ("line", lines[i])):
                    _, C, H, W = shp
                    f = smooth_field(seed, stream, C, H, W) * scale
                    if white:
                        f = f + white * hash_normal(seed, stream + 5000, C * H * W).reshape(C, H, W)
                    out[f"{kind}_{what}_{g}.{i}"] = f.reshape(shp).astype(np.float32)
                    stream += 1

    def linear(name: str, n_out: int, n_in: int, bias: bool, st: int, zero_bias: bool = False):
        bound = 1.0 / math.sqrt(n_in)  # nn.Linear default init range
        w = (hash_uniform(seed, st, n_out * n_in).reshape(n_out, n_in) * 2 - 1) * bound * mlp_gain
        out[f"{name}.weight"] = w.astype(np.float32)
        if bias:
            b = np.zeros(n_out) if zero_bias else (hash_uniform(seed, st + 1, n_out) * 2 - 1) * bound
            out[f"{name}.bias"] = b.astype(np.float32)

    linear("basis_mat_yin", cfg.app_dim, sum(cfg.app_n_comp), False, 9000)
    linear("basis_mat_yang", cfg.app_dim, sum(cfg.app_n_comp), False, 9010)
    if cfg.shadingMode != "RGB":   # RGBRender is a plain function: no parameters (tensorBase.py:37-39)
        linear("renderModule.mlp.0", cfg.featureC, cfg.in_mlpC, True, 9020)
        linear("renderModule.mlp.2", cfg.featureC, cfg.featureC, True, 9030)
        linear("renderModule.mlp.4", 3, cfg.featureC, True, 9040, zero_bias=True)  # tensorBase.py:66
    if cfg.use_envmap:
        h = cfg.envmap_res_H
        out["envmap.emission"] = (smooth_field(seed, 9100, 3, 2 * h, h, lattice=24) * 1.5).astype(np.float32)
    return out


def carve_empty_space(weights: Dict[str, np.ndarray], cfg: SceneConfig, r_keep=((0.14, 0.30), (0.54, 0.68)), phi_cut=(0.0, 0.28),
                       density_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """A copy of `weights` whose DENSITY field has real empty space, like a trained scene: the density feature is exactly 0 outside
    the radial shells `r_keep` and inside the phi wedge `phi_cut` (fractions of the r / phi index ranges; both grids alike), so that
    sigma = softplus(density_shift) there and the reference's alpha-mask rule (EgoNeRF.py:437-489: lattice alpha >= 1e-4 after a 3^3
    max-pool) marks those voxels empty.  Zeroing the factors of every VM term does it exactly: for r outside the shells planes 0, 1
    (x = r) and line 2 (r); inside the wedge planes 1, 2 (y = phi) and line 0 (phi).  Bilinear taps make the transition one texel
    wide.  The appearance field is untouched (the reference's mask acts on sigma only)."""
    n_r, _n_th, n_ph = cfg.grid
    r_idx = np.arange(n_r) / max(n_r - 1, 1)
    keep_r = np.zeros(n_r, bool)
    for lo, hi in r_keep:
        keep_r |= (r_idx >= lo) & (r_idx < hi)
    ph_idx = np.arange(n_ph) / max(n_ph - 1, 1)
    keep_ph = ~((ph_idx >= phi_cut[0]) & (ph_idx < phi_cut[1]))
    out = dict(weights)
    for g in ("yin", "yang"):
        p0 = weights[f"density_plane_{g}.0"].copy() * density_gain   # (1, C, N_theta, N_r)
        p1 = weights[f"density_plane_{g}.1"].copy() * density_gain   # (1, C, N_phi, N_r)
        p2 = weights[f"density_plane_{g}.2"].copy() * density_gain   # (1, C, N_phi, N_theta)
        l0 = weights[f"density_line_{g}.0"].copy()                   # (1, C, N_phi, 1)
        l2 = weights[f"density_line_{g}.2"].copy()                   # (1, C, N_r, 1)
        p0[..., ~keep_r] = 0
        p1[..., ~keep_r] = 0
        l2[:, :, ~keep_r, :] = 0
        p1[:, :, ~keep_ph, :] = 0
        p2[:, :, ~keep_ph, :] = 0
        l0[:, :, ~keep_ph, :] = 0
        out.update({f"density_plane_{g}.0": p0, f"density_plane_{g}.1": p1, f"density_plane_{g}.2": p2,
                    f"density_line_{g}.0": l0, f"density_line_{g}.2": l2})
    return out


def white_envmap(seed: int, h: int) -> np.ndarray:
    """[3, 2h, h] fp32 emission map of independent uniforms in [-3, 3) (pre-sigmoid), for index-exact envmap tests."""
    return ((hash_uniform(seed, 7, 3 * 2 * h * h) * 6 - 3).reshape(3, 2 * h, h)).astype(np.float32)


def psnr_target(ref_rgb: np.ndarray, seed: int = 77, target_db: float = 30.0) -> np.ndarray:
    """A synthetic ground truth at ~`target_db` PSNR from a rendered image: clamp(ref + sigma * N(0,1), 0, 1), integer-hash noise.
    Lets a test state the north_star's PSNR clause directly: PSNR(candidate, gt) - PSNR(reference, gt) on a realistic ~30 dB target."""
    ref = np.asarray(ref_rgb, np.float64)
    sigma = 10.0 ** (-target_db / 20.0)
    return np.clip(ref + sigma * hash_normal(seed, 11, ref.size).reshape(ref.shape), 0.0, 1.0)


def delta_psnr(candidate_rgb, reference_rgb, seed: int = 77, target_db: float = 30.0) -> Tuple[float, float, float]:
    """(PSNR(candidate, gt) - PSNR(reference, gt), PSNR(candidate, gt), PSNR(reference, gt)) in dB, float64, with
    gt = psnr_target(reference) and PSNR = -10 log10(mean((img - gt)^2)) as renderer.py:156-157 computes it."""
    cand, ref = np.asarray(candidate_rgb, np.float64), np.asarray(reference_rgb, np.float64)
    gt = psnr_target(ref, seed, target_db)
    p_c = -10.0 * np.log10(np.mean((cand - gt) ** 2))
    p_r = -10.0 * np.log10(np.mean((ref - gt) ** 2))
    return float(p_c - p_r), float(p_c), float(p_r)


def make_rays(n: int, seed: int = 1, origin_extent: float = 0.25) -> np.ndarray:
    """[n,6] fp32: o ~ U(-e,e)^3, d = normalised approx-normal vector (sqrt is IEEE-exact)."""
    o = (hash_uniform(seed, 1, n * 3).reshape(n, 3) * 2 - 1) * origin_extent
    d = hash_normal(seed, 2, n * 3).reshape(n, 3)
    nrm = np.sqrt((d * d).sum(-1, keepdims=True))
    d = d / np.maximum(nrm, 1e-12)
    return np.concatenate([o, d], -1).astype(np.float32)


def erp_rays(H: int, W: int, row0: int = 0, row1: int | None = None, origin=(0.0, 0.0, 0.0)) -> np.ndarray:
    """Equirectangular ray layout of dataLoader/ray_utils.py:24-40 (identity pose), rows [row0,row1).

    Uses libm sin/cos (numpy float32, like the reference's torch float32), so these rays are
    box-local inputs, not golden data.
    """
    row1 = H if row1 is None else row1
    i = np.arange(W, dtype=np.float32)[None, :] + np.float32(0.5)
    j = np.arange(row0, row1, dtype=np.float32)[:, None] + np.float32(0.5)
    phi = (1 - 2 * i / W) * np.float32(np.pi)
    theta = (1 - 2 * j / H) * np.float32(np.pi / 2)
    d = np.stack(np.broadcast_arrays(-np.cos(theta) * np.sin(phi), np.sin(theta) * np.ones_like(phi),
                                     -np.cos(theta) * np.cos(phi)), -1).reshape(-1, 3)
    o = np.broadcast_to(np.asarray(origin, np.float32), d.shape)
    return np.concatenate([o, d], -1).astype(np.float32)


def build_coords(cfg: "SceneConfig", device):
    """The scene's YinYangSphericalCoords exactly as train.py:118-130 constructs it."""
    from .coordinates import YinYangSphericalCoords
    return YinYangSphericalCoords(device, cfg.aabb, exp_r=True, N_voxel=cfg.n_voxel, r0=cfg.r0, interval_th=cfg.interval_th)


def build_model(cfg: "SceneConfig", weights, device="cuda"):
    """egonerf_amd EgoNeRF for a synthetic scene, with the reference's ctor kwargs (train.py:163-171 resolved values) and
    `weights` (reference state_dict layout, e.g. make_weights) loaded."""
    import torch
    from .model import EgoNeRF
    coords = build_coords(cfg, device)
    assert coords.resolution == cfg.grid
    model = EgoNeRF(torch.from_numpy(cfg.aabb), cfg.grid, device, coords, density_n_comp=list(cfg.density_n_comp),
                    appearance_n_comp=list(cfg.app_n_comp), app_dim=cfg.app_dim, near_far=[cfg.near, cfg.far],
                    shadingMode=cfg.shadingMode, alphaMask_thres=1e-4, density_shift=cfg.density_shift,
                    distance_scale=cfg.distance_scale, pos_pe=6, view_pe=cfg.view_pe, fea_pe=cfg.fea_pe, featureC=cfg.featureC,
                    step_ratio=0.5, fea2denseAct="softplus", use_envmap=cfg.use_envmap, envmap_res_H=cfg.envmap_res_H,
                    coarse_sigma_grid_update_rule="conv", coarse_sigma_grid_reso=None, interval_th=cfg.interval_th)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items() if k != "envmap.emission"})
    if cfg.use_envmap:
        model.envmap.load_envmap(weights["envmap.emission"], device=device)
    if torch.device(device).type == "cuda":
        model.update_coarse_sigma_grid()
    model.eval()
    return model

