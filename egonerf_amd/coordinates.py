"""Yin-yang spherical coordinates: host-side mirror of models/coordinates.py (YinYangSphericalCoords
over GenericSphericalCoords) for the configuration every shipped EgoNeRF config uses
(`coordinates = yinyang`, `exp_r`, `interval_th`; configs/EgoNeRF/common.txt:2,15).

The host only builds the small float32 constants (centre, angular ranges, max radius, the
linearised-exponential radius LUT and the sample schedule) with the same torch float32 operations the
reference uses, so they are bit-identical; the per-point work (from_cartesian, normalize_coord) runs in
libegonerf_hip.so.
"""
from __future__ import annotations

from math import exp, log, pi, sqrt

import torch

from . import _lib


def linearised_exp_grid(r0: float, ratio, n: int) -> torch.Tensor:
    """Shell radii r[0]=0, r[i]=r0*ratio**(i-1), with every shell thinner than r0 replaced by an
    arithmetic run of step r0 (tail shifted to stay continuous); float32 on the CPU.

    Restates extra/test_exp_r.py:10-15 + models/EgoNeRF.py:71-76 (Python-float ratio) and
    models/coordinates.py:117-124 (0-dim float32 tensor ratio).  The float32 roundings matter
    (see tests/golden/stages.npz `sched/*`, `normr/*`), so dtype promotion is kept as in the reference:
    r0 * n_lin is a float32 tensor product, not a Python double.
    """
    idx = torch.arange(n)
    r = torch.zeros(n, dtype=torch.float32)
    expo = (idx[1:] - 1) if isinstance(ratio, torch.Tensor) else (idx[1:].float() - 1)
    r[1:] = r0 * ratio ** expo
    step = r[1:] - r[:-1]
    run = torch.cumsum(step, 0)
    n_lin = (step <= r0).sum()  # 0-dim int64 tensor on purpose (float32 product below)
    r[n_lin + 1:] = r[n_lin + 1:] + r0 * n_lin - run[n_lin - 1]
    r[: n_lin + 1] = torch.arange(n_lin + 1) * r0
    return r


class YinYangSphericalCoords:
    """[r_n, theta_n, phi_n, r_e, theta_e, phi_e, Y]: Y=0 yin grid, Y=1 yang grid (coordinates.py:432-438)."""

    def __init__(self, device, aabb, exp_r=True, N_voxel=None, r0=None, interval_th=False):
        self.device = device
        aabb = torch.as_tensor(aabb, dtype=torch.float32).cpu()
        self.aabb = aabb
        self.center = aabb.sum(0).div(2)  # coordinates.py:76
        self.exp_r = exp_r
        self.interval_th = interval_th
        self.update_aabb(aabb)
        self.set_resolution(self.N_to_reso(N_voxel, aabb), r0=r0)

    _REFERENCE_ATTRS = ("center", "device", "near", "far", "inv_diff", "exp_r", "interval_th", "N_r", "N_theta", "N_phi", "r0", "ratio")

    def __reduce_ex__(self, protocol):
        """Inside compat.reference_pickle_paths() (EgoNeRF.save) the object pickles as the reference's
        models.coordinates.YinYangSphericalCoords with exactly its attribute set (no device LUT cache, no aabb copy)."""
        from . import compat
        if not compat.pickling_as_reference():
            return super().__reduce_ex__(protocol)
        state = {k: getattr(self, k) for k in self._REFERENCE_ATTRS if hasattr(self, k)}
        state["device"] = str(state["device"])
        return compat.reduce_as_reference("models.coordinates", "YinYangSphericalCoords", state)

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_lut_dev"] = None  # device tensors: rebuilt on demand
        return state

    def __setstate__(self, state):
        """Also accepts the attribute set of a pickled reference object (models.coordinates.YinYangSphericalCoords inside
        a `.th` checkpoint's kwargs, tensorBase.py:264): device, center, near, far, inv_diff, exp_r, interval_th, N_*, r0,
        ratio, resolution."""
        self.__dict__.update(state)
        self.__dict__.setdefault("aabb", None)
        self._lut_dev = None
        for k in ("center", "near", "far", "inv_diff"):
            if torch.is_tensor(self.__dict__.get(k)):
                self.__dict__[k] = self.__dict__[k].detach().cpu().float()
        if torch.is_tensor(self.__dict__.get("ratio")):
            self.ratio = self.ratio.detach().cpu()
        if not hasattr(self, "resolution"):
            self.resolution = [self.N_r, self.N_theta, self.N_phi]

    # -- constants ---------------------------------------------------------------------------------
    def _get_max_r(self, aabb) -> torch.Tensor:
        """Distance centre -> farthest aabb corner, float32 (coordinates.py:187-204)."""
        lo, hi = torch.as_tensor(aabb).tolist()
        corners = torch.tensor([[lo[b] if (i >> b) & 1 else hi[b] for b in range(3)] for i in range(8)])
        return (corners - self.center).pow(2).sum(1).sqrt().amax()

    def update_aabb(self, new_aabb):
        """coordinates.py:500-505."""
        max_r = self._get_max_r(new_aabb)
        self.near = torch.tensor([0, pi / 4, -3 * pi / 4, 0, pi / 4, -3 * pi / 4])
        self.far = torch.stack([max_r, torch.tensor(3 * pi / 4), torch.tensor(3 * pi / 4)] * 2)
        self.inv_diff = 1.0 / (self.far - self.near)
        self._lut_dev = None

    def N_to_reso(self, n_voxels, bbox=None):
        """coordinates.py:507-520 (keeps Python's inexact pow(x, 1/3): 27e6 -> N_r starts from 149)."""
        n_r = int(pow(n_voxels, 1 / 3) / 2)
        n_t = int(n_r * 2 * sqrt(3) / 3)
        n_p = n_t * 3
        even = lambda v: v + 1 if v % 2 else v
        return [even(n_r), even(n_t), even(n_p)]

    def set_resolution(self, resolution, r0=None):
        """coordinates.py:206-215."""
        self.N_r, self.N_theta, self.N_phi = resolution
        self.resolution = list(resolution)
        if self.exp_r:
            self.r0 = r0 if r0 is not None else 0.05
            self.ratio = pow(self.far[0] / self.r0, 1 / (self.N_r - 1))
        self._lut_dev = None

    def reference_r_grid(self, downsample=None) -> torch.Tensor:
        """Knots of the piecewise-linear radius normalisation, r -> (cell + fraction) / (len - 1).
        interval_th: the [N_r+1] LUT the reference rebuilds on every normalize_r call (coordinates.py:116-124; `downsample` is
        ignored there).  Plain exponential grid (coordinates.py:132-155): [0, r0, r0 ratio, ..., r0 ratio^(N-1)] with
        N = N_r // downsample and ratio = (far / r0)^(1 / (N - 1)); the reference finds the cell with a truncated logarithm,
        which is the same piecewise-linear map."""
        if self.interval_th:
            ratio = pow(self.far[0] / self.r0, 1 / (self.N_r - 1))
            return linearised_exp_grid(self.r0, ratio, self.N_r + 1)
        n = self.N_r if downsample is None else self.N_r // downsample
        ratio = self.ratio if downsample is None else pow(self.far[0] / self.r0, 1 / (n - 1))
        ratio = torch.as_tensor(ratio, dtype=torch.float32).cpu()
        knots = self.r0 * torch.pow(ratio, torch.arange(n))
        return torch.cat([torch.zeros(1), knots]).float()

    def sample_schedule(self, near: float, far: float, n_samples: int) -> torch.Tensor:
        """Radial sample offsets r[S] of EgoNeRF.sample_ray_exp: interval_th branch (EgoNeRF.py:69-76) or the plain
        exponential schedule (EgoNeRF.py:59-67)."""
        if not self.interval_th:
            ratio = 1 + (pi / 2.0) / n_samples
            r0 = (far - near) * (ratio - 1) / (pow(ratio, n_samples) - 1)
            rng = torch.arange(n_samples)[None].float()
            return (torch.pow(ratio, rng) @ torch.tril(torch.ones(n_samples, n_samples), diagonal=-1).T * r0)[0]
        ratio = exp(log((far - near) / self.r0) / (n_samples - 1))
        return linearised_exp_grid(self.r0, ratio, n_samples)

    def upsample_axis_positions(self, res_target, axis: int) -> torch.Tensor:
        """Normalised [-1, 1] positions, on the CURRENT grid, of the samples of a `res_target` grid along `axis`
        (coordinates.py:226-266 for r: the new shell radii located in the old LUT; :27-39 for the angles: linspace)."""
        n = int(res_target[axis])
        if axis != 0:
            return torch.linspace(-1, 1, n)
        self._require_supported()
        ratio = pow(self.far[0] / self.r0, 1 / (n - 1))  # coordinates.py:238 (0-dim tensor pow)
        if self.interval_th:
            new = linearised_exp_grid(self.r0, ratio, n)
        else:   # plain exponential grid (coordinates.py:260-262): 0, r0, r0 ratio, ..., r0 ratio^(n-2)
            new = torch.cat([torch.zeros(1), self.r0 * torch.pow(torch.as_tensor(ratio, dtype=torch.float32).cpu(), torch.arange(n - 1))]).float()
        G = self.reference_r_grid()
        k_out = torch.clamp(torch.searchsorted(G, new, right=True), 1, G.shape[0] - 1)
        k_in = k_out - 1
        r01 = (k_in + (new - G[k_in]) / (G[k_out] - G[k_in])) / self.N_r  # normalize_r, coordinates.py:125-131
        return r01 * 2 - 1

    @_lib.device_guard
    def up_sampling_VM(self, weights: torch.Tensor, res_target, ids):
        """coordinates.py:226-266: `weights` (1, C, H, W) channel-last table; ids = [axis of H, axis of W] (plane) or
        [axis] (line) -> resampled channel-last nn.Parameter."""
        from . import _lib
        assert len(ids) in (1, 2), "ids should be 1 or 2!"
        if not weights.is_cuda:
            raise RuntimeError("up_sampling_VM: needs a HIP device tensor (the EgoNeRF path has no CPU fallback)")
        _, C_, H, W = weights.shape
        src = weights.detach()
        if not src.permute(0, 2, 3, 1).is_contiguous():
            src = src.contiguous(memory_format=torch.channels_last)
        dev = weights.device
        ys = self.upsample_axis_positions(res_target, ids[0]).to(dev, torch.float32).contiguous()
        xs = (self.upsample_axis_positions(res_target, ids[1]) if len(ids) == 2 else -torch.ones(1)).to(dev, torch.float32).contiguous()
        dst = torch.empty(1, ys.numel(), xs.numel(), C_, device=dev)
        _lib.check(_lib.load().ego_resample_table(src.data_ptr(), C_, H, W, xs.data_ptr(), ys.data_ptr(), ys.numel(), xs.numel(),
                                                  dst.data_ptr(), _lib.stream_handle()), "ego_resample_table")
        return torch.nn.Parameter(dst.permute(0, 3, 1, 2))

    def _require_supported(self):
        if not self.exp_r:
            raise NotImplementedError("the HIP path covers exponential r grids (exp_r; every shipped EgoNeRF config); "
                                      "uniform r grids are out of scope")

    def lut_device(self, device, downsample=None) -> torch.Tensor:
        if not isinstance(self._lut_dev, dict):
            self._lut_dev = {}
        key = (str(torch.device(device)), None if self.interval_th else downsample)
        if key not in self._lut_dev:
            self._lut_dev[key] = self.reference_r_grid(downsample).to(device)
        return self._lut_dev[key]

    def fill_scene(self, sc: "_lib.Scene", device, downsample=2) -> None:
        """Writes the coordinate block of the C-ABI scene struct.  `downsample` = what the first (or only) pass of the render
        normalises r with (EgoNeRF.py:524 passes 2; it only matters for the plain exponential grid, whose fine pass after
        resampling then uses the full grid, EgoNeRF.py:546 -> r_lut_fine)."""
        self._require_supported()
        lut = self.lut_device(device, downsample)
        sc.center[:] = self.center.tolist()
        sc.ang_near[:] = self.near[1:3].tolist()
        sc.ang_inv[:] = self.inv_diff[1:3].tolist()
        sc.r_lut = lut.data_ptr()
        sc.n_r_lut = lut.numel()
        sc.n_r = lut.numel() - 1
        if not self.interval_th and downsample is not None:
            fine = self.lut_device(device, None)
            sc.r_lut_fine, sc.n_r_lut_fine, sc.n_r_fine = fine.data_ptr(), fine.numel(), fine.numel() - 1

    def _scene(self, device, downsample=2):
        sc = _lib.new_scene()
        self.fill_scene(sc, device, downsample)
        return sc

    # -- per-point ops (HIP) ---------------------------------------------------------------------------
    @_lib.device_guard
    def from_cartesian(self, xyz_points: torch.Tensor) -> torch.Tensor:
        """[...,3] -> [...,7] (coordinates.py:468-498)."""
        _require_cuda(xyz_points, "from_cartesian")
        if xyz_points.shape[-1] != 3:
            raise IndexError(f"from_cartesian: expected [..., 3], got {tuple(xyz_points.shape)}")
        x = xyz_points.contiguous().float()
        out = torch.empty(*x.shape[:-1], 7, device=x.device, dtype=torch.float32)
        sc = self._scene(x.device)
        _lib.check(_lib.load().ego_from_cartesian(sc, x.data_ptr(), x.numel() // 3, out.data_ptr(), _lib.stream_handle()),
                   "ego_from_cartesian")
        return out

    @_lib.device_guard
    def normalize_coord(self, unnormalized_coords: torch.Tensor, downsample=None) -> torch.Tensor:
        """[...,7] -> [...,7] in [-1,1] (+flag) (coordinates.py:442-466).  `downsample` is ignored by the interval_th grid
        (coordinates.py:112-117) and coarsens the plain exponential one (coordinates.py:137-139)."""
        _require_cuda(unnormalized_coords, "normalize_coord")
        if unnormalized_coords.shape[-1] != 7:
            raise IndexError(f"normalize_coord: expected [..., 7], got {tuple(unnormalized_coords.shape)}")
        x = unnormalized_coords.contiguous().float()
        out = torch.empty_like(x)
        sc = self._scene(x.device, downsample)
        _lib.check(_lib.load().ego_normalize_coord(sc, x.data_ptr(), x.numel() // 7, out.data_ptr(), _lib.stream_handle()),
                   "ego_normalize_coord")
        return out


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the EgoNeRF hot path runs only on the HIP device "
                           "(there is no CPU fallback)")
