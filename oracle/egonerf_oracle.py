"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of EgoNeRF's volume-rendering hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
The product package (`egonerf_amd/`) never does: its hot path is the HIP library and it raises if
that library is missing.

Parity status: PINNED.  `oracle/capture_golden.py` imports the real reference (read-only, this
container only) and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every stage of
this file against those vectors (per-stage intermediates on a tiny grid, outputs on the full grid).

It deliberately keeps the reference's ATen op sequence (F.grid_sample on (1,C,H,W) tables, boolean
mask gather/scatter per yin/yang grid, cat -> Linear, cumprod, searchsorted, sort), because it doubles
as the "reference PyTorch CPU path" timed by bench.py (`cpu_baseline.kind == "port"`).
`dtype=torch.float64` evaluates the same maths in double (LUTs stay float32-quantised like the
reference's) and is used by tests to decide which of two fp32 answers is closer to the truth.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

MAT_MODE = ((0, 1), (0, 2), (1, 2))  # models/EgoNeRF.py:30-33 (same for yin and yang)
VEC_MODE = (2, 1, 0)
GRIDS = ("yin", "yang")


def linearised_exp_grid(r0: float, ratio, n: int) -> torch.Tensor:
    """r[0]=0, r[i]=r0*ratio^(i-1) in float32, then every shell thinner than r0 is replaced by an
    arithmetic run of step r0 and the tail shifted to stay continuous.

    extra/test_exp_r.py:10-15 (index2r, hard float32) + models/EgoNeRF.py:71-76 (ratio = Python
    float) / models/coordinates.py:118-124 (ratio = 0-dim float32 tensor).
    """
    idx = torch.arange(n)
    r = torch.zeros(n, dtype=torch.float32)
    if isinstance(ratio, torch.Tensor):
        r[1:] = r0 * ratio ** (idx[1:] - 1)
    else:
        r[1:] = r0 * ratio ** (idx[1:].float() - 1)
    step = r[1:] - r[:-1]
    csum = torch.cumsum(step, 0)
    n_lin = (step <= r0).sum()  # stays a 0-dim int64 tensor: r0 * n_lin must round in float32
    r[: n_lin + 1] = torch.arange(n_lin + 1) * r0
    r[n_lin + 1:] = r[n_lin + 1:] + r0 * n_lin - csum[n_lin - 1]
    return r


class OracleScene:
    """Holds reference-layout weights + resolved scalars and evaluates the path on CPU."""

    def __init__(self, cfg, weights: Dict[str, np.ndarray], dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        self.w = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in weights.items()}
        self.grid = list(cfg.grid)
        self.aabb = torch.as_tensor(cfg.aabb)
        self.center = self.aabb.sum(0).div(2)  # coordinates.py:76
        # coordinates.py:187-204 (_get_max_r) and :500-505 (update_aabb), float32 like the reference
        lo, hi = self.aabb.tolist()
        corners = torch.tensor([[lo[b] if (i >> b) & 1 else hi[b] for b in range(3)] for i in range(8)])
        self.far_r = (corners - self.center).pow(2).sum(1).sqrt().amax()
        pi = math.pi
        self.ang_near = torch.tensor([pi / 4, -3 * pi / 4])
        ang_far = torch.tensor([3 * pi / 4, 3 * pi / 4])
        self.ang_inv = 1.0 / (ang_far - self.ang_near)
        self.set_resolution(self.grid, r0=cfg.r0)
        self.coarse = None
        self.update_coarse_sigma_grid()
        # opt-in skipping (the reference's EgoNeRF.forward has none; TensorBase.forward, tensorBase.py:464-487, defines it):
        self.alpha_mask = None   # (vol_yin, vol_yang) float {0,1} volumes (1,1,N_phi,N_theta,N_r), TensorBase.forward semantics
        self.term_eps = 0.0      # early termination: weight := 0 where the incoming transmittance < term_eps
        self.weight_thres = None  # TensorBase.forward's rayMarch_weight_thres appearance skip (None = off, as in EgoNeRF.forward)

    def set_resolution(self, resolution, r0=None):
        """coordinates.py:206-215.  Quirk kept: without an explicit r0 (as train.py:377 calls it after an upsample) the
        radial knee resets to 0.05 whatever the scene's config said."""
        self.grid = [int(v) for v in resolution]
        self.r0 = r0 if r0 is not None else 0.05
        ratio = pow(self.far_r / self.r0, 1 / (self.grid[0] - 1))  # coordinates.py:117 (tensor pow)
        self.r_lut = linearised_exp_grid(self.r0, ratio, self.grid[0] + 1)  # coordinates.py:118-124

    # ---- parameters -------------------------------------------------------------------------
    def table(self, kind: str, what: str, g: str, i: int) -> torch.Tensor:
        return self.w[f"{kind}_{what}_{g}.{i}"]

    def update_coarse_sigma_grid(self):
        """2x average-pooled density tables (EgoNeRF.py:124-131)."""
        self.coarse = {}
        for g in GRIDS:
            for i in range(3):
                self.coarse[f"plane_{g}.{i}"] = F.avg_pool2d(self.table("density", "plane", g, i), 2, 2)
                self.coarse[f"line_{g}.{i}"] = F.avg_pool1d(self.table("density", "line", g, i).squeeze(-1), 2, 2).unsqueeze(-1)

    # ---- row A: sample schedule -------------------------------------------------------------
    def sample_schedule(self, S: int) -> torch.Tensor:
        """Radial offsets r[S] (float32) of EgoNeRF.sample_ray_exp: interval_th branch (EgoNeRF.py:69-76) or the plain
        exponential one (EgoNeRF.py:59-67: exclusive prefix sums of r0' ratio^k, ratio = 1 + (pi/2)/S)."""
        c = self.cfg
        if not getattr(c, "interval_th", True):
            ratio = 1 + (math.pi / 2.0) / S
            r0 = (c.far - c.near) * (ratio - 1) / (pow(ratio, S) - 1)
            rng = torch.arange(S)[None].float()
            return (torch.pow(ratio, rng) @ torch.tril(torch.ones(S, S), diagonal=-1).T * r0)[0]
        ratio = math.exp(math.log((c.far - c.near) / self.r0) / (S - 1))
        return linearised_exp_grid(self.r0, ratio, S)

    def sample_ray(self, rays_o, rays_d, S: int, jitter: Optional[torch.Tensor] = None, step_ratio: float = 0.5):
        """tensorBase.py:308-327 (exp_sampling=False): aabb entry distance clamped to [near, far], then uniform steps of
        stepSize = mean(aabbSize / (gridSize - 1)) * step_ratio (tensorBase.py:206-217); train: + U[0,1) steps."""
        c = self.cfg
        aabb = self.aabb.to(rays_o.dtype)
        step = ((aabb[1] - aabb[0]) / (torch.tensor(self.grid, dtype=torch.float32) - 1)).mean() * step_ratio
        vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
        t_min = torch.minimum((aabb[1] - rays_o) / vec, (aabb[0] - rays_o) / vec).amax(-1).clamp(min=c.near, max=c.far)
        rng = torch.arange(S)[None].float()
        if jitter is not None:
            rng = rng.repeat(rays_d.shape[-2], 1) + jitter
        z = t_min[..., None] + step.to(rays_o.dtype) * rng.to(rays_o.dtype)
        return rays_o[..., None, :] + rays_d[..., None, :] * z[..., None], z

    def sample_ray_exp(self, rays_o, rays_d, S: int, jitter: Optional[torch.Tensor] = None):
        """EgoNeRF.py:56-87.  `jitter` [N,S] in [0,1) replaces torch.rand_like for is_train."""
        c = self.cfg
        if jitter is not None and not getattr(c, "interval_th", True):
            # plain exponential schedule in training: the noise goes into the exponent (EgoNeRF.py:63-67)
            ratio = 1 + (math.pi / 2.0) / S
            r0 = (c.far - c.near) * (ratio - 1) / (pow(ratio, S) - 1)
            rng = torch.arange(S)[None].float().repeat(rays_d.shape[-2], 1) + jitter.float()
            z = (c.near + torch.pow(ratio, rng) @ torch.tril(torch.ones(S, S), diagonal=-1).T * r0).to(self.dtype)
            return rays_o[..., None, :] + rays_d[..., None, :] * z[..., None], z
        r = self.sample_schedule(S).repeat(rays_d.shape[-2], 1)
        if jitter is not None:
            step = r[:, 1:] - r[:, :-1]
            step = torch.cat([step, step[:, -1:]], -1)
            r = r + step * jitter
        z = (self.cfg.near + r).to(self.dtype)
        xyz = rays_o[..., None, :] + rays_d[..., None, :] * z[..., None]
        return xyz, z

    # ---- row B: Cartesian -> yin-yang 7-vector ----------------------------------------------
    def from_cartesian(self, xyz: torch.Tensor, grid_choice: Optional[torch.Tensor] = None) -> torch.Tensor:
        """coordinates.py:468-498: [r,th,ph,0,0,0,0] (yin) or [0,0,0,r,th_e,ph_e,1] (yang).
        `grid_choice` (tests only; same leading shape, 0 = yin, 1 = yang) overrides the inclusive border test: a point within
        an ulp of a region border lands on either grid depending on the libm's acos/atan2 rounding (the reference itself
        chooses differently on CPU and GPU), and the two grids hold independent tables."""
        pi = math.pi
        d = xyz - self.center.to(xyz.dtype)
        r = d.pow(2).sum(-1).sqrt()
        th_n = torch.acos(d[..., 2] / r).nan_to_num_()
        ph_n = torch.atan2(d[..., 1], d[..., 0])
        yin = (pi / 4 <= th_n) & (th_n <= 3 * pi / 4) & (-3 * pi / 4 <= ph_n) & (ph_n <= 3 * pi / 4)
        if grid_choice is not None:
            yin = grid_choice.to(yin.device) == 0
        th_e = torch.acos(d[..., 1] / r).nan_to_num_()
        ph_e = torch.atan2(d[..., 2], -d[..., 0])
        zero = torch.zeros_like(r)
        as_yin = torch.stack([r, th_n, ph_n, zero, zero, zero, zero], -1)
        as_yang = torch.stack([zero, zero, zero, r, th_e, ph_e, torch.ones_like(r)], -1)
        return torch.where(yin[..., None], as_yin, as_yang)

    def yin_margin(self, xyz: torch.Tensor) -> torch.Tensor:
        """Signed angular margin (radians, float64) of the yin/yang decision of from_cartesian: > 0 inside the yin region,
        < 0 outside; |margin| below a few float32 ulps of pi (~5e-7) means the grid choice depends on the libm."""
        pi = math.pi
        d = xyz.double() - self.center.double()
        r = d.pow(2).sum(-1).sqrt()
        th = torch.acos((d[..., 2] / r).clamp(-1, 1)).nan_to_num()
        ph = torch.atan2(d[..., 1], d[..., 0])
        return torch.stack([th - pi / 4, 3 * pi / 4 - th, ph + 3 * pi / 4, 3 * pi / 4 - ph], -1).amin(-1)

    # ---- row C: normalisation ----------------------------------------------------------------
    def normalize_r(self, r: torch.Tensor, downsample=None) -> torch.Tensor:
        """coordinates.py:110-131,156 (interval_th branch; `downsample` is ignored there) or :132-156 (plain exponential grid:
        cell index from a logarithm truncated through an int cast, `downsample` coarsens the grid)."""
        n_r = self.grid[0]
        if not getattr(self.cfg, "interval_th", True):
            r0 = self.r0
            if downsample is None:
                ratio = pow(self.far_r / r0, 1 / (n_r - 1))
            else:
                n_r = n_r // downsample
                ratio = pow(self.far_r / r0, 1 / (n_r - 1))
            k = (torch.log(r / r0) / torch.log(ratio.to(r.dtype))).to(torch.int32)
            small = r < r0
            r_in = torch.where(small, torch.zeros_like(r), r0 * torch.pow(ratio.to(r.dtype), k))
            r_out = torch.where(small, torch.full_like(r, r0), r0 * torch.pow(ratio.to(r.dtype), k + 1))
            lin = (r - r_in) / (r_out - r_in)
            return torch.where(small, r / r0, 1 + k + lin) / n_r
        G = self.r_lut.to(r.dtype)
        k_out = torch.clamp(torch.searchsorted(G, r.contiguous(), side="right"), 1, G.shape[0] - 1)
        k_in = k_out - 1
        frac = (r - G[k_in]) / (G[k_out] - G[k_in])
        return (k_in + frac) / n_r

    def normalize_coord(self, c7: torch.Tensor, downsample=2) -> torch.Tensor:
        """coordinates.py:442-466 (exp_r branch); EgoNeRF.forward always passes downsample=2 (EgoNeRF.py:524)."""
        near = self.ang_near.to(c7.dtype)
        inv = self.ang_inv.to(c7.dtype)
        parts = []
        for base in (0, 3):
            parts.append((self.normalize_r(c7[..., base], downsample) * 2 - 1).unsqueeze(-1))
            parts.append((c7[..., base + 1: base + 3] - near) * inv * 2 - 1)
        parts.append(c7[..., 6:7])
        return torch.cat(parts, -1)

    # ---- rows D, D', F: VM lookups -----------------------------------------------------------
    @staticmethod
    def _vm_taps(planes, lines, p3: torch.Tensor):
        """Per plane/line pair: bilinear plane sample [C,M] and linear line sample [C,M]
        (F.grid_sample, align_corners=True, zero padding; EgoNeRF.py:301-346)."""
        outs = []
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            gp = torch.stack([p3[:, m0], p3[:, m1]], -1).view(1, -1, 1, 2)
            gl = torch.stack([torch.zeros_like(p3[:, 0]), p3[:, VEC_MODE[i]]], -1).view(1, -1, 1, 2)
            P = F.grid_sample(planes[i], gp, align_corners=True).view(planes[i].shape[1], -1)
            L = F.grid_sample(lines[i], gl, align_corners=True).view(lines[i].shape[1], -1)
            outs.append((P, L))
        return outs

    def density_feature(self, c7n: torch.Tensor, coarse: bool = False) -> torch.Tensor:
        """EgoNeRF.py:291-347 (full-res) / :232-289 (pooled tables): sum_i relu(sum_c P_ic L_ic)."""
        flat = c7n.reshape(-1, 7)
        out = torch.zeros(flat.shape[0], dtype=flat.dtype)
        is_yin = flat[:, -1] == 0
        for g, sel, base in (("yin", is_yin, 0), ("yang", ~is_yin, 3)):
            if not bool(sel.any()):
                continue
            p3 = flat[sel][:, base: base + 3]
            if coarse:
                planes = [self.coarse[f"plane_{g}.{i}"] for i in range(3)]
                lines = [self.coarse[f"line_{g}.{i}"] for i in range(3)]
            else:
                planes = [self.table("density", "plane", g, i) for i in range(3)]
                lines = [self.table("density", "line", g, i) for i in range(3)]
            acc = torch.zeros(p3.shape[0], dtype=flat.dtype)
            for P, L in self._vm_taps(planes, lines, p3):
                acc = acc + F.relu((P * L).sum(0))
            out[sel] = acc
        return out.view(c7n.shape[:-1])

    def app_feature(self, c7n: torch.Tensor) -> torch.Tensor:
        """EgoNeRF.py:349-413: concat of 3x48 products -> per-grid Linear(144->27, no bias)."""
        flat = c7n.reshape(-1, 7)
        out = torch.zeros(flat.shape[0], self.cfg.app_dim, dtype=flat.dtype)
        is_yin = flat[:, -1] == 0
        for g, sel, base in (("yin", is_yin, 0), ("yang", ~is_yin, 3)):
            if not bool(sel.any()):
                continue
            p3 = flat[sel][:, base: base + 3]
            planes = [self.table("app", "plane", g, i) for i in range(3)]
            lines = [self.table("app", "line", g, i) for i in range(3)]
            taps = self._vm_taps(planes, lines, p3)
            Pc = torch.cat([t[0] for t in taps])
            Lc = torch.cat([t[1] for t in taps])
            out[sel] = F.linear((Pc * Lc).T, self.w[f"basis_mat_{g}.weight"])
        return out.view(*c7n.shape[:-1], self.cfg.app_dim)

    # ---- row E ---------------------------------------------------------------------------------
    def feature2density(self, f: torch.Tensor) -> torch.Tensor:
        """tensorBase.py:415-419 (softplus, torch threshold 20)."""
        return F.softplus(f + self.cfg.density_shift)

    @staticmethod
    def raw2alpha(sigma: torch.Tensor, dist: torch.Tensor):
        """tensorBase.py:22-27."""
        alpha = 1.0 - torch.exp(-sigma * dist)
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)
        return alpha, alpha * T[:, :-1], T[:, -1:]

    # ---- row G ---------------------------------------------------------------------------------
    @staticmethod
    def positional_encoding(x: torch.Tensor, n_freq: int) -> torch.Tensor:
        """tensorBase.py:14-19: element-major / frequency-minor, all sines then all cosines."""
        bands = (2 ** torch.arange(n_freq).float()).to(x.dtype)
        p = (x[..., None] * bands).reshape(x.shape[:-1] + (n_freq * x.shape[-1],))
        return torch.cat([torch.sin(p), torch.cos(p)], -1)

    def mlp_fea(self, viewdirs: torch.Tensor, feat: torch.Tensor) -> torch.Tensor:
        """MLPRender_Fea.forward (tensorBase.py:68-78); 150 -> 128 -> 128 -> 3, sigmoid.  shadingMode "MLP" = MLPRender.forward
        (tensorBase.py:121-129): the same network without the feature encoding; "RGB" = RGBRender (tensorBase.py:37-39): the features."""
        c = self.cfg
        if getattr(c, "shadingMode", "MLP_Fea") == "RGB":
            return feat
        x = torch.cat([feat, viewdirs, self.positional_encoding(feat, getattr(c, "head_fea_pe", c.fea_pe)),
                       self.positional_encoding(viewdirs, c.view_pe)], -1)
        h = F.relu(F.linear(x, self.w["renderModule.mlp.0.weight"], self.w["renderModule.mlp.0.bias"]))
        h = F.relu(F.linear(h, self.w["renderModule.mlp.2.weight"], self.w["renderModule.mlp.2.bias"]))
        return torch.sigmoid(F.linear(h, self.w["renderModule.mlp.4.weight"], self.w["renderModule.mlp.4.bias"]))

    # ---- row I ---------------------------------------------------------------------------------
    @staticmethod
    def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n: int, u: Optional[torch.Tensor] = None):
        """dataLoader/ray_utils.py:156-187.  `u` [N,n] replaces torch.rand for is_train."""
        w = weights + 1e-5
        pdf = w / w.sum(-1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
        if u is None:
            u = torch.linspace(0.0, 1.0, steps=n, dtype=cdf.dtype).expand(list(cdf.shape[:-1]) + [n])
        u = u.contiguous()
        idx = torch.searchsorted(cdf.detach(), u, right=True)
        lo = (idx - 1).clamp(min=0)
        hi = idx.clamp(max=cdf.shape[-1] - 1)
        c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
        b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
        den = c_hi - c_lo
        den = torch.where(den < 1e-5, torch.ones_like(den), den)
        return b_lo + (u - c_lo) / den * (b_hi - b_lo)

    # ---- row M ---------------------------------------------------------------------------------
    def sample_alpha(self, c7n: torch.Tensor) -> torch.Tensor:
        """YinYangAlphaGridMask.sample_alpha (EgoNeRF.py:19-24): trilinear grid_sample of the {0,1} volumes."""
        flat = c7n.reshape(-1, 7)
        out = torch.empty(flat.shape[0], dtype=flat.dtype)
        is_yin = flat[:, -1] == 0
        for vol, sel, base in ((self.alpha_mask[0], is_yin, 0), (self.alpha_mask[1], ~is_yin, 3)):
            if bool(sel.any()):
                out[sel] = F.grid_sample(vol.to(flat.dtype), flat[sel][:, base:base + 3].view(1, -1, 1, 1, 3), align_corners=True).view(-1)
        return out.view(c7n.shape[:-1])

    def build_alpha_mask(self, step_size: float, thres: float = 1e-4):
        """EgoNeRF.getDenseAlpha + updateAlphaMask (EgoNeRF.py:437-489)."""
        g = self.grid
        lin = [torch.linspace(0, 1, n) for n in g]
        norm = torch.stack(torch.meshgrid(*lin, indexing="ij"), -1) * 2 - 1
        zeros3, flag = torch.zeros_like(norm), torch.zeros_like(norm[..., :1])
        vols = []
        for c7 in (torch.cat([norm, zeros3, flag], -1), torch.cat([zeros3, norm, flag + 1], -1)):
            a = 1 - torch.exp(-self.feature2density(self.density_feature(c7.view(-1, 7))) * step_size)
            a = a.view(g).clamp(0, 1).transpose(0, 2).contiguous()[None, None]
            a = F.max_pool3d(a, kernel_size=3, padding=1, stride=1).view(g[::-1])
            vols.append((a >= thres).float()[None, None])
        self.alpha_mask = (vols[0], vols[1])
        return self.alpha_mask

    # ---- row J ---------------------------------------------------------------------------------
    def envmap_radiance(self, dirs: torch.Tensor) -> torch.Tensor:
        """models/envmap.py:6-14,26-34: u=(d_z+1)/2 on the h-wide axis, v=(atan2(d_y,d_x)+pi)/2pi."""
        d = F.normalize(dirs, dim=-1)
        u = (d[:, 2] + 1) * 0.5
        v = (torch.atan2(d[:, 1], d[:, 0]) + math.pi) / (2 * math.pi)
        uv = torch.stack([u, v], 1) * 2 - 1
        em = self.w["envmap.emission"]
        s = F.grid_sample(em[None], uv[None, :, None, :], align_corners=True)
        return torch.sigmoid(s.permute(0, 2, 3, 1).reshape(-1, 3))

    # ---- rows A..J assembled: EgoNeRF.forward ----------------------------------------------------
    def forward(self, rays: torch.Tensor, n_coarse: int, n_fine: int = 0, resampling: bool = False,
                use_coarse_sample: bool = True, is_train: bool = False,
                jitter: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None,
                keep: bool = False, exp_sampling: bool = True, grid_choice: Optional[torch.Tensor] = None):
        """EgoNeRF.forward (EgoNeRF.py:491-602), exp_sampling + interval_th path.

        Returns (rgb[N,3], depth[N], bg|None, env|None, alpha[N,S(+1)]) and, if keep, a dict of
        intermediates.
        """
        rays = rays.to(self.dtype)
        c = self.cfg
        o, viewdirs = rays[:, :3], rays[:, 3:6]
        if is_train and jitter is None:
            jitter = torch.rand(rays.shape[0], n_coarse)
        sampler = self.sample_ray_exp if exp_sampling else self.sample_ray  # EgoNeRF.py:506-513
        xyz, z = sampler(o, viewdirs, n_coarse, jitter if is_train else None)
        if not is_train:
            z = z[0].repeat(xyz.shape[0], 1)  # EgoNeRF.py:515-516 (ray 0's distances for every ray, also with sample_ray)
        dists = z[..., 1:] - z[..., :-1]
        dists = torch.cat([dists, dists[..., -1:]], -1)
        c7 = self.from_cartesian(xyz, None if resampling else grid_choice)  # `grid_choice` addresses the rendered samples
        c7n = self.normalize_coord(c7)
        inter = dict(z_coarse=z, c7=c7, c7n=c7n, xyz_coarse=xyz) if keep else None

        if resampling:
            sf = self.density_feature(c7n, coarse=True)
            _, cw, _ = self.raw2alpha(self.feature2density(sf), dists * c.distance_scale)
            z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
            if is_train and u is None:
                u = torch.rand(rays.shape[0], n_fine)
            z_new = self.sample_pdf(z_mid, cw[..., 1:-1], n_fine, u if is_train else None).detach()
            if use_coarse_sample:
                z, _ = torch.sort(torch.cat([z, z_new], -1), -1)
            else:
                z, _ = torch.sort(z_new, -1)
            dists = z[..., 1:] - z[..., :-1]
            dists = torch.cat([dists, dists[..., -1:]], -1)
            xyz = o[:, None, :] + viewdirs[:, None, :] * z[..., None]
            c7n = self.normalize_coord(self.from_cartesian(xyz, grid_choice), downsample=None)  # the fine pass: full grid (EgoNeRF.py:546)
            if keep:
                inter.update(coarse_sigma_feat=sf, coarse_weight=cw, z_new=z_new, z_fine=z, xyz_fine=xyz)

        sf = self.density_feature(c7n)
        sigma = self.feature2density(sf)
        af = self.app_feature(c7n)
        vd = viewdirs.view(-1, 1, 3).expand(xyz.shape)
        rgb = self.mlp_fea(vd.reshape(-1, 3), af.reshape(-1, c.app_dim)).view(*xyz.shape[:2], 3)
        # mask application + appearance skip with TensorBase.forward's semantics (pinned to the reference through
        # tensorbase_skip_composite / tests/golden/skip_semantics.npz); both off = EgoNeRF.forward (EgoNeRF.py:579-598)
        mask_alpha = self.sample_alpha(c7n) if self.alpha_mask is not None else None
        rgb_map, depth, alpha, weight, bg_w, app_mask = tensorbase_skip_composite(sigma, mask_alpha, None, dists, z, rgb, rays[..., -1],
                                                                           c.distance_scale, self.weight_thres)
        if mask_alpha is not None:
            sigma = torch.where(mask_alpha > 0, sigma, torch.zeros_like(sigma))
        if self.term_eps > 0:  # product extension (no reference counterpart): weights behind transmittance < eps are dropped
            T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
            weight = torch.where(T < self.term_eps, torch.zeros_like(weight), weight)
            rgb_map = (weight[..., None] * torch.where(app_mask[..., None], rgb, torch.zeros_like(rgb))).sum(-2)
            depth = (weight * z).sum(-1) + (1.0 - weight.sum(-1)) * rays[..., -1]
        acc = weight.sum(-1)
        bg_map = env_map = None
        if c.use_envmap:
            alpha = torch.cat([alpha, torch.ones_like(alpha[..., :1])], -1)
            env_map = self.envmap_radiance(viewdirs)
            bg_map = bg_w * env_map
            rgb_map = rgb_map + bg_map
        rgb_map = rgb_map.clamp(0, 1)
        depth = depth.detach()  # EgoNeRF.py:595-598 (no_grad; d_z quirk)
        if keep:
            inter.update(sigma_feat=sf, sigma=sigma, weight=weight, bg_weight=bg_w, app_feat=af,
                         rgb_samples=rgb, z=z, acc=acc)
            return (rgb_map, depth, bg_map, env_map, alpha), inter
        return rgb_map, depth, bg_map, env_map, alpha


    # ---- training-step extras (train.py:245-330) ---------------------------------------------------
    @staticmethod
    def tv_loss(x: torch.Tensor) -> torch.Tensor:
        """utils.py:155-171 (TVLoss, weight 1): 2 * (sum dH^2 / count_h + sum dW^2 / count_w) / batch."""
        _, C, H, W = x.shape
        h_tv = (x[:, :, 1:, :] - x[:, :, :-1, :]).pow(2).sum()
        w_tv = (x[:, :, :, 1:] - x[:, :, :, :-1]).pow(2).sum()
        return 2 * (h_tv / (C * (H - 1) * W) + w_tv / (C * H * (W - 1))) / x.shape[0]

    def TV_loss(self, kind: str) -> torch.Tensor:
        """EgoNeRF.py:214-228: planes only, 1e-2 each, yin and yang."""
        total = 0
        for i in range(3):
            for g in GRIDS:
                total = total + self.tv_loss(self.table(kind, "plane", g, i)) * 1e-2
        return total

    def density_L1(self) -> torch.Tensor:
        """EgoNeRF.py:206-212."""
        total = 0
        for i in range(3):
            for g in GRIDS:
                total = total + self.table("density", "plane", g, i).abs().mean() + self.table("density", "line", g, i).abs().mean()
        return total

    def vector_comp_diffs(self) -> torch.Tensor:
        """EgoNeRF.py:189-201: mean |off-diagonal| of the line Gram matrices, density + app, yin + yang."""
        total = 0
        for kind in ("density", "app"):
            for g in GRIDS:
                for i in range(3):
                    v = self.table(kind, "line", g, i)
                    v = v.reshape(v.shape[1], v.shape[2])
                    gram = v @ v.T
                    off = gram[~torch.eye(gram.shape[0], dtype=torch.bool)]
                    total = total + off.abs().mean()
        return total

    def upsample_volume_grid(self, res_target):
        """EgoNeRF.py:415-435 + coordinates.py:226-266 (exp_r, interval_th) + :27-39: bilinear (align_corners) resample of
        every table; angular axes at linspace(-1,1), the radial axis at the *new* shell radii located in the *old* grid.
        The caller then calls set_resolution (train.py:376-377)."""
        c = self.cfg
        n_new = int(res_target[0])
        ratio = pow(self.far_r / self.r0, 1 / (n_new - 1))  # coordinates.py:238: far tensor -> 0-dim tensor ratio
        if getattr(c, "interval_th", True):
            new_r = linearised_exp_grid(self.r0, ratio, n_new)
        else:   # plain exponential grid, coordinates.py:260-262
            new_r = torch.cat([torch.zeros(1), self.r0 * torch.pow(torch.as_tensor(ratio, dtype=torch.float32), torch.arange(n_new - 1))]).float()
        r_samples = self.normalize_r(new_r) * 2 - 1  # positions on the old grid
        axis = lambda a: r_samples if a == 0 else torch.linspace(-1, 1, int(res_target[a]))
        for kind in ("density", "app"):
            for g in GRIDS:
                for i in range(3):
                    ax, ay = MAT_MODE[i]  # plane (1,C,size[ay],size[ax]) sampled at (x = ax, y = ay)
                    xs, ys = axis(ax).to(self.dtype), axis(ay).to(self.dtype)
                    grid = torch.stack(torch.meshgrid(ys, xs, indexing="ij")[::-1], -1)[None]
                    key = f"{kind}_plane_{g}.{i}"
                    self.w[key] = F.grid_sample(self.w[key].detach(), grid, align_corners=True)
                    ls = axis(VEC_MODE[i]).to(self.dtype)
                    grid = torch.stack([-torch.ones_like(ls), ls], -1)[None, :, None, :]
                    key = f"{kind}_line_{g}.{i}"
                    self.w[key] = F.grid_sample(self.w[key].detach(), grid, align_corners=True)
        self.update_coarse_sigma_grid()


def ray_entropy_loss(alpha: torch.Tensor) -> torch.Tensor:
    """utils.py:175-183."""
    p = alpha / (alpha.sum(-1, keepdim=True) + 1e-10)
    return (-(p * torch.log2(p + 1e-10)).sum(-1)).mean()


def tensorbase_skip_composite(sigma_dense: torch.Tensor, mask_alpha: Optional[torch.Tensor], ray_valid: Optional[torch.Tensor],
                              dists: torch.Tensor, z: torch.Tensor, rgb_dense: torch.Tensor, rays_last: torch.Tensor,
                              distance_scale: float, weight_thres: Optional[float]):
    """The skip logic of TensorBase.forward (models/tensorBase.py:464-507), on dense per-sample inputs:
      * :464-469  a sample is kept iff it is inside the aabb (`ray_valid`) and its alpha-mask lookup is > 0;
      * :471-478  sigma = 0 for every other sample (its density is never evaluated);
      * :480      alpha / weight / bg_weight = raw2alpha(sigma, dists * distance_scale);
      * :482-487  colour = 0 for samples with weight <= rayMarch_weight_thres (appearance + MLP never evaluated);
      * :489-490,503-507  acc = sum w, rgb = sum w c, depth = sum w z + (1 - acc) * rays[..., -1]   (weights are NOT zeroed).
    sigma_dense [N,S] = feature2density of every sample; mask_alpha [N,S] or None (no mask); ray_valid [N,S] bool or None
    (EgoNeRF.forward ignores the aabb flags, EgoNeRF.py:85); weight_thres None = no appearance skip.
    -> (rgb_map before the background / clamp, depth, alpha, weight, bg_weight, app_mask)."""
    keep = torch.ones_like(sigma_dense, dtype=torch.bool) if ray_valid is None else ray_valid.clone()
    if mask_alpha is not None:
        keep &= mask_alpha > 0
    sigma = torch.where(keep, sigma_dense, torch.zeros_like(sigma_dense))
    alpha, weight, bg_w = OracleScene.raw2alpha(sigma, dists * distance_scale)
    app_mask = torch.ones_like(keep) if weight_thres is None else weight > weight_thres
    rgb = torch.where(app_mask[..., None], rgb_dense, torch.zeros_like(rgb_dense))
    acc = weight.sum(-1)
    rgb_map = (weight[..., None] * rgb).sum(-2)
    depth = (weight * z).sum(-1) + (1.0 - acc) * rays_last
    return rgb_map, depth, alpha, weight, bg_w, app_mask


def erp_rays_reference(H: int, W: int, c2w: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """[H*W, 6] rays of an equirectangular camera: dataLoader/ray_utils.py:24-40 (get_ray_directions_360: pixel centres,
    phi = (1 - 2i/W) pi, theta = (1 - 2j/H) pi/2, d = [-cos(theta) sin(phi), sin(theta), -cos(theta) cos(phi)]), the datasets'
    normalisation (dataset_egocentric_video.py:57-58 / dataset_omniblender.py:42-43) and :85-113 (get_rays: d @ R^T, o = t)."""
    i = torch.tile(torch.arange(W), (H, 1)) + 0.5
    j = torch.tile(torch.arange(H), (W, 1)).T + 0.5
    phi = (1 - 2 * i / W) * np.pi
    theta = (1 - 2 * j / H) * np.pi / 2
    d = torch.stack([-torch.cos(theta) * torch.sin(phi), torch.sin(theta), -torch.cos(theta) * torch.cos(phi)], -1)
    if normalize:
        d = d / torch.norm(d, dim=-1, keepdim=True)
    rays_d = d @ c2w[:3, :3].T
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return torch.cat([rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)], 1)


def volume_render(scene: OracleScene, rays: torch.Tensor, chunk: int = 4096, **kw):
    """Chunk loop of renderer.py:11-79 (torch outputs, no D2H variant)."""
    outs = [scene.forward(rays[i:i + chunk], **kw) for i in range(0, rays.shape[0], chunk)]
    cat = lambda j: None if outs[0][j] is None else torch.cat([o[j] for o in outs])
    return cat(0), cat(1), cat(2), cat(3), cat(4)


def psnr(img: torch.Tensor, gt: torch.Tensor) -> float:
    """renderer.py:156-157."""
    return float(-10.0 * np.log(torch.mean((img - gt) ** 2).item()) / np.log(10.0))


def rgb_ssim(img0, img1, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, return_map=False):
    """utils.py:104-152: per-channel 'valid' separable Gaussian statistics in float64 over float32 images (squares and
    products are taken in float32 first, as the reference does with torch tensors), clipped variances, mean of the map."""
    a, b = np.asarray(img0, np.float32), np.asarray(img1, np.float32)
    hw = filter_size // 2
    shift = (2 * hw - filter_size + 1) / 2
    filt = np.exp(-0.5 * ((np.arange(filter_size) - hw + shift) / filter_sigma) ** 2)
    filt /= filt.sum()

    def blur(z):
        z = z.astype(np.float64)
        v = sum(filt[k] * z[k: z.shape[0] - filter_size + 1 + k] for k in range(filter_size))
        return sum(filt[k] * v[:, k: v.shape[1] - filter_size + 1 + k] for k in range(filter_size))

    mu0, mu1 = blur(a), blur(b)
    mu00, mu11, mu01 = mu0 * mu0, mu1 * mu1, mu0 * mu1
    s00 = np.maximum(0.0, blur(a * a) - mu00)
    s11 = np.maximum(0.0, blur(b * b) - mu11)
    s01 = blur(a * b) - mu01
    s01 = np.sign(s01) * np.minimum(np.sqrt(s00 * s11), np.abs(s01))
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    ssim_map = ((2 * mu01 + c1) * (2 * s01 + c2)) / ((mu00 + mu11 + c1) * (s00 + s11 + c2))
    return ssim_map if return_map else float(np.mean(ssim_map))



def ws_rows(n: int) -> np.ndarray:
    """extra/ws_ssim.py:12-24 (generate_ws / estws): the weight of row i of an n-row equirectangular map, float64."""
    return np.array([np.cos((i + 0.5 - n / 2) * np.pi / n) for i in range(n)])


def ws_mean(smap: np.ndarray) -> float:
    """extra/ws_ssim.py:29-31: sum(map * ws) / sum(ws), ws = row weights broadcast over the columns."""
    ws = np.repeat(ws_rows(smap.shape[0])[:, None], smap.shape[1], 1)
    return float(np.sum(smap * ws) / ws.sum())


def ws_psnr(img: np.ndarray, gt: np.ndarray) -> float:
    """Latitude-weighted PSNR (the WS-PSNR of the 360-video literature with extra/ws_ssim.py's weights), float64."""
    w = ws_rows(img.shape[0])[:, None, None]
    d = np.asarray(img, np.float64) - np.asarray(gt, np.float64)
    return float(10 * np.log10(1.0 / ((d * d * w).sum() / (w.sum() * img.shape[1] * img.shape[2]))))


def sh_render(viewdirs: torch.Tensor, features: torch.Tensor) -> torch.Tensor:
    """models/tensorBase.py:30-34 with the degree-2 bases of models/sh.py:87-112."""
    x, y, z = viewdirs.unbind(-1)
    c1, c2 = 0.4886025119029199, (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
    Y = torch.stack([torch.full_like(x, 0.28209479177387814), -c1 * y, c1 * z, -c1 * x, c2[0] * x * y, c2[1] * y * z,
                     c2[2] * (2.0 * z * z - x * x - y * y), c2[3] * x * z, c2[4] * (x * x - y * y)], -1)
    return torch.relu((Y[:, None] * features.view(-1, 3, 9)).sum(-1) + 0.5)
