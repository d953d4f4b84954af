"""Host-side mirror of the reference model surface for the volume-rendering path:

    models/tensorBase.py : raw2alpha, MLPRender_Fea, TensorBase (ctor kwargs, feature2density, save/load)
    models/EgoNeRF.py    : EgoNeRF (forward, compute_densityfeature, compute_coarse_densityfeature,
                           compute_appfeature, sample_ray_exp, update_coarse_sigma_grid,
                           get_optparam_groups, save, load)
    models/envmap.py     : EnvironmentMap

Same names, argument meaning, state-dict keys and return tuples, so the parity tests read like calls into
the reference.  All per-sample arithmetic runs in libegonerf_hip.so through the C ABI (egonerf_amd/_lib.py);
PyTorch only owns device memory, streams and parameters.  There is no CPU fallback: calling the path with
CPU tensors raises.

Parameters keep the reference's shapes ((1,C,H,W) planes, (1,C,L,1) lines) but are allocated channel-last
([H][W][C] in memory), which is the layout the kernels gather from; `state_dict()` keys and shapes are the
reference's, so checkpoints interchange.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import math
import numpy as np
import torch
import torch.nn

from . import _lib
from .coordinates import YinYangSphericalCoords, _require_cuda

MAT_MODE = [[0, 1], [0, 2], [1, 2]]  # models/EgoNeRF.py:30-33
VEC_MODE = [2, 1, 0]


def _call(name: str, *args) -> None:
    _lib.check(getattr(_lib.load(), name)(*args), name)


def _last_dim(t: torch.Tensor, n: int, what: str) -> None:
    """The reference indexes `[..., k]` and fails with an IndexError on a short last axis; a raw-pointer call would silently
    reinterpret the memory instead."""
    if t.dim() < 1 or t.shape[-1] != n:
        raise IndexError(f"{what}: expected [..., {n}], got {tuple(t.shape)}")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().float()


def _channel_last_param(C_: int, H: int, W: int, scale: float, device) -> torch.nn.Parameter:
    """Reference-shaped (1,C,H,W) parameter whose memory is [H][W][C]."""
    mem = torch.empty(1, H, W, C_, device=device, dtype=torch.float32)
    mem.copy_(scale * torch.randn(1, C_, H, W).permute(0, 2, 3, 1))
    return torch.nn.Parameter(mem.permute(0, 3, 1, 2))


def _carve_channel_last(shapes, device, dtype=torch.float32) -> List[torch.Tensor]:
    """One flat allocation carved into [1,H,W,C]-memory tables, returned as reference-shaped (1,C,H,W) views.  The tables of a
    field come from one buffer so that the kernels can address every tap as `base + 32-bit byte offset` (csrc/ego_device.h
    DevField; the library rejects a field whose tables span 4 GB or more).  Each table starts on a 256-byte boundary."""
    esz = torch.empty((), dtype=dtype).element_size()
    align = 256 // esz
    offs, total = [], 0
    for (C_, H, W) in shapes:
        offs.append(total)
        total += (C_ * H * W + align - 1) // align * align
    flat = torch.empty(total, device=device, dtype=dtype)
    return [flat[o:o + C_ * H * W].view(1, H, W, C_).permute(0, 3, 1, 2) for o, (C_, H, W) in zip(offs, shapes)]


def _table_ptr(t: torch.Tensor) -> int:
    """Device pointer of a (1,C,H,W) tensor after checking its memory really is [H][W][C]."""
    if not t.permute(0, 2, 3, 1).is_contiguous():
        raise RuntimeError(f"table of shape {tuple(t.shape)} / strides {t.stride()} is not channel-last; "
                           "assign parameters with param.data.copy_(...) so the layout is kept")
    return t.data_ptr()


# ---------------------------------------------------------------------------------------------------
# models/tensorBase.py
# ---------------------------------------------------------------------------------------------------
@_lib.device_guard
def raw2alpha(sigma: torch.Tensor, dist: torch.Tensor):
    """alpha, weights, bg_weight of tensorBase.py:22-27 (sigma, dist: [N_rays, N_samples])."""
    _require_cuda(sigma, "raw2alpha")
    sigma, dist = _f32c(sigma), _f32c(dist)
    N, S = sigma.shape
    alpha, weight = torch.empty_like(sigma), torch.empty_like(sigma)
    bg = torch.empty(N, 1, device=sigma.device)
    _call("ego_raw2alpha", sigma.data_ptr(), dist.data_ptr(), N, S, alpha.data_ptr(), weight.data_ptr(), bg.data_ptr(),
          _lib.stream_handle())
    return alpha, weight, bg


class EnvironmentMap:
    """models/envmap.py:17-37: plain requires_grad tensor `emission` [3, 2h, h] (not an nn.Parameter)."""

    def __init__(self, h=1000, init_strategy="random", device="cuda"):
        if init_strategy == "random":
            self.emission = torch.rand((3, 2 * h, h), requires_grad=True, device=device)
        elif init_strategy == "zero":
            self.emission = torch.zeros((3, 2 * h, h), requires_grad=True, device=device)
        else:
            raise ValueError("Unknown environment map initialization: {}".format(init_strategy))

    @_lib.device_guard
    def get_radiance(self, direction: torch.Tensor) -> torch.Tensor:
        _require_cuda(direction, "EnvironmentMap.get_radiance")
        d = _f32c(direction)
        if torch.is_grad_enabled() and self.emission.requires_grad:  # envmap pre-training (train.py:218-236)
            from .train import EnvRadianceFunction
            return EnvRadianceFunction.apply(self.emission, d)
        out = torch.empty(d.shape[0], 3, device=d.device)
        sc = _lib.new_scene()
        em = self.emission.detach().contiguous()
        sc.envmap, sc.envmap_h = em.data_ptr(), em.shape[2]
        _call("ego_envmap_radiance", sc, d.data_ptr(), d.shape[0], out.data_ptr(), _lib.stream_handle())
        return out

    def __reduce_ex__(self, protocol):
        """Inside compat.reference_pickle_paths() (EgoNeRF.save): pickles as models.envmap.EnvironmentMap {emission}."""
        from . import compat
        if not compat.pickling_as_reference():
            return super().__reduce_ex__(protocol)
        return compat.reduce_as_reference("models.envmap", "EnvironmentMap", {"emission": self.emission})

    def load_envmap(self, emission, device):
        self.emission = torch.tensor(np.asarray(emission.detach().cpu() if torch.is_tensor(emission) else emission),
                                     requires_grad=True, device=device, dtype=torch.float32)


@_lib.device_guard
def SHRender(xyz_sampled, viewdirs: torch.Tensor, features: torch.Tensor) -> torch.Tensor:
    """models/tensorBase.py:30-34: degree-2 SH colour head, viewdirs [M,3], features [M,27] -> rgb [M,3] = relu(SH . f + 0.5).
    (`xyz_sampled` is unused, as in the reference.  The reference's EgoNeRF.forward cannot run with this head — it hands
    SHRender [N,S,3] view directions against [N*S,3,9] features and the broadcast raises — so it is offered as the stage
    function only.)"""
    _require_cuda(viewdirs, "SHRender")
    d, f = _f32c(viewdirs.reshape(-1, 3)), _f32c(features.reshape(-1, 27))
    if d.shape[0] != f.shape[0]:
        raise RuntimeError("SHRender: viewdirs and features disagree on the number of points")
    out = torch.empty(d.shape[0], 3, device=d.device)
    _call("ego_sh_render", d.data_ptr(), f.data_ptr(), d.shape[0], out.data_ptr(), _lib.stream_handle())
    return out


class MLPRender_Fea(torch.nn.Module):
    """tensorBase.py:54-78.  The Linear layers only hold the weights (state-dict keys `mlp.{0,2,4}.*`);
    forward runs the fused PE + 150->128->128->3 + sigmoid MFMA kernel."""

    def __init__(self, inChannel, viewpe=6, feape=6, featureC=128):
        super().__init__()
        self.in_mlpC = 2 * viewpe * 3 + 2 * feape * inChannel + 3 + inChannel
        self.viewpe, self.feape = viewpe, feape
        self.mlp = torch.nn.Sequential(torch.nn.Linear(self.in_mlpC, featureC), torch.nn.ReLU(inplace=True),
                                       torch.nn.Linear(featureC, featureC), torch.nn.ReLU(inplace=True),
                                       torch.nn.Linear(featureC, 3))
        torch.nn.init.constant_(self.mlp[-1].bias, 0)
        self._owner = None  # set by the model: provides the packed weights

    @_lib.device_guard
    def forward(self, pts, viewdirs, features):
        _require_cuda(features, "MLPRender_Fea")
        if self._owner is None:
            raise RuntimeError("MLPRender_Fea must belong to an EgoNeRF model (it owns the packed MFMA weights)")
        model = self._owner()
        v = _f32c(viewdirs.reshape(-1, 3))
        f = _f32c(features.reshape(-1, features.shape[-1]))
        out = torch.empty(f.shape[0], 3, device=f.device)
        _call("ego_mlp_fea", model.scene(), v.data_ptr(), f.data_ptr(), f.shape[0], out.data_ptr(), _lib.stream_handle())
        return out.view(*features.shape[:-1], 3)


class MLPRender(MLPRender_Fea):
    """tensorBase.py:107-129 (shadingMode 'MLP'): mlp_in = [features, viewdirs, PE(viewdirs)] - MLPRender_Fea without the feature
    encoding, same state-dict keys `mlp.{0,2,4}.*`; runs the any-shape kernels (csrc/ego_generic.hip) with fea_pe = 0."""

    def __init__(self, inChannel, viewpe=6, featureC=128):
        super().__init__(inChannel, viewpe, 0, featureC)
        del self.feape   # the reference class has no such attribute


def RGBRender(xyz_sampled, viewdirs, features):
    """tensorBase.py:37-39 (shadingMode 'RGB', app_dim == 3): the colour is the appearance feature.  A plain function like the
    reference's (no parameters: `isinstance(renderModule, torch.nn.Module)` is False, EgoNeRF.py:152)."""
    return features


class TensorBase(torch.nn.Module):
    """Constructor surface and shared helpers of tensorBase.py:132-186, 206-217, 241-295, 415-419."""

    def __init__(self, aabb, gridSize, device, coordinates, density_n_comp=8, appearance_n_comp=24, app_dim=27,
                 shadingMode="MLP_PE", alphaMask=None, near_far=[2.0, 6.0], density_shift=-10, alphaMask_thres=0.001,
                 distance_scale=25, rayMarch_weight_thres=0.0001, pos_pe=6, view_pe=6, fea_pe=6, featureC=128,
                 step_ratio=2.0, fea2denseAct="softplus", use_envmap=False, envmap_res_H=1000, envmap=None,
                 coarse_sigma_grid_update_rule=None, coarse_sigma_grid_reso=None, interval_th=False):
        super().__init__()
        as_list = lambda n: list(n) if isinstance(n, (list, tuple)) else [n] * 3
        self.density_n_comp, self.app_n_comp, self.app_dim = as_list(density_n_comp), as_list(appearance_n_comp), app_dim
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32)
        self.alphaMask = alphaMask
        self.device = device
        self.density_shift, self.alphaMask_thres, self.distance_scale = density_shift, alphaMask_thres, distance_scale
        self.rayMarch_weight_thres, self.fea2denseAct = rayMarch_weight_thres, fea2denseAct
        self.near_far, self.step_ratio = list(near_far), step_ratio
        self.update_stepSize(gridSize)
        self.envmap = None
        if use_envmap:
            if envmap is None:
                self.init_envmap(envmap_res_H, init_strategy="random", device=device)
            else:
                self.envmap = EnvironmentMap(h=envmap.emission.shape[2], init_strategy="zero", device=device)
                self.envmap.load_envmap(envmap.emission, device=device)
        self.shadingMode, self.pos_pe, self.view_pe, self.fea_pe, self.featureC = shadingMode, pos_pe, view_pe, fea_pe, featureC
        self.init_render_func(shadingMode, pos_pe, view_pe, fea_pe, featureC, device)
        self.coordinates = coordinates
        self.coarse_sigma_grid_update_rule = coarse_sigma_grid_update_rule

    def init_render_func(self, shadingMode, pos_pe, view_pe, fea_pe, featureC, device):
        """tensorBase.py:186-200, for the heads the reference's EgoNeRF.forward can run with: 'MLP_Fea' (every shipped config,
        configs/EgoNeRF/common.txt:34), 'MLP' and 'RGB'.  'MLP_PE' (the ctor default) and 'SH' raise inside the reference's
        EgoNeRF.forward - MLPRender_PE is handed 7-column yin-yang coordinates for a 3-column position encoding (shape error in its
        first Linear), SHRender [N,S,3] directions against [N*S,3,9] features - so they are refused here at construction
        (SHRender exists as a stage function)."""
        if shadingMode == "MLP_Fea":
            self.renderModule = MLPRender_Fea(self.app_dim, view_pe, fea_pe, featureC).to(device)
        elif shadingMode == "MLP":
            self.renderModule = MLPRender(self.app_dim, view_pe, featureC).to(device)
        elif shadingMode == "RGB":
            assert self.app_dim == 3   # tensorBase.py:198
            self.renderModule = RGBRender
        else:
            raise NotImplementedError(f"shadingMode {shadingMode!r}: EgoNeRF.forward runs with 'MLP_Fea', 'MLP' or 'RGB' (the reference's own "
                                      "forward raises with 'MLP_PE' and 'SH')")

    @property
    def head_fea_pe(self) -> int:
        """Feature-encoding frequencies of the head as built (MLPRender ignores the `fea_pe` argument, RGBRender has no network)."""
        return self.fea_pe if self.shadingMode == "MLP_Fea" else 0

    @property
    def head_in_mlpC(self) -> int:
        return self.renderModule.in_mlpC if self.shadingMode != "RGB" else 0

    @property
    def head_hidden(self) -> int:
        return self.featureC if self.shadingMode != "RGB" else 0

    def init_envmap(self, envmap_res_H, init_strategy="zero", device="cuda"):
        self.envmap = EnvironmentMap(h=envmap_res_H, init_strategy=init_strategy, device=device)

    def update_stepSize(self, gridSize):
        """tensorBase.py:206-217 (values unused by EgoNeRF.forward, kept for get_kwargs/alpha-mask parity)."""
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invaabbSize = 2.0 / self.aabbSize
        self.gridSize = torch.LongTensor(list(gridSize))
        self.units = self.aabbSize / (self.gridSize - 1)
        self.stepSize = torch.mean(self.units) * self.step_ratio
        self.aabbHalfDiag = torch.sqrt(torch.sum(torch.square(self.aabbSize))) / 2.0
        self.nSamples = int((self.aabbHalfDiag / self.stepSize).item()) + 1

    def get_kwargs(self):
        return {"aabb": self.aabb, "gridSize": self.gridSize.tolist(), "density_n_comp": self.density_n_comp,
                "appearance_n_comp": self.app_n_comp, "app_dim": self.app_dim, "density_shift": self.density_shift,
                "alphaMask_thres": self.alphaMask_thres, "distance_scale": self.distance_scale,
                "rayMarch_weight_thres": self.rayMarch_weight_thres, "fea2denseAct": self.fea2denseAct,
                "near_far": self.near_far, "step_ratio": self.step_ratio, "shadingMode": self.shadingMode,
                "pos_pe": self.pos_pe, "view_pe": self.view_pe, "fea_pe": self.fea_pe, "featureC": self.featureC,
                "coordinates": self.coordinates, "use_envmap": self.envmap is not None, "envmap": self.envmap,
                "coarse_sigma_grid_update_rule": self.coarse_sigma_grid_update_rule}

    @_lib.device_guard
    def feature2density(self, density_features: torch.Tensor) -> torch.Tensor:
        """tensorBase.py:415-419."""
        _require_cuda(density_features, "feature2density")
        f = _f32c(density_features)
        out = torch.empty_like(f)
        sc = _lib.new_scene()
        sc.act_softplus, sc.density_shift = int(self.fea2denseAct == "softplus"), float(self.density_shift)
        _call("ego_feature2density", sc, f.data_ptr(), f.numel(), out.data_ptr(), _lib.stream_handle())
        return out


# ---------------------------------------------------------------------------------------------------
# models/EgoNeRF.py
# ---------------------------------------------------------------------------------------------------
class YinYangAlphaGridMask(torch.nn.Module):
    """models/EgoNeRF.py:11-24: two {0,1} volumes shaped (1,1,N_phi,N_theta,N_r); sample_alpha = trilinear lookup."""

    def __init__(self, device, alpha_volume_yin, alpha_volume_yang):
        super().__init__()
        self.device = device
        self.alpha_volume_yin = alpha_volume_yin.view(1, 1, *alpha_volume_yin.shape[-3:])
        self.alpha_volume_yang = alpha_volume_yang.view(1, 1, *alpha_volume_yang.shape[-3:])
        self._bytes = self._cells = None

    def packed(self) -> torch.Tensor:
        """[2][N_phi][N_theta][N_r] uint8 (what the kernels read)."""
        if self._bytes is None:
            self._bytes = torch.stack([self.alpha_volume_yin[0, 0], self.alpha_volume_yang[0, 0]]).gt(0).to(torch.uint8).contiguous()
        return self._bytes

    def cell_or(self) -> Optional[torch.Tensor]:
        """[2][N_phi-1][N_theta-1][N_r-1] uint8: OR of every cell's eight corner voxels (ego_scene.occ_cell: the march's one-byte
        answer to "trilinear mask value > 0" for samples strictly inside a cell)."""
        vol = self.packed()
        if min(vol.shape[1:]) < 2:
            return None
        if self._cells is None:
            self._cells = torch.nn.functional.max_pool3d(vol.float()[None], kernel_size=2, stride=1)[0].to(torch.uint8).contiguous()
        return self._cells

    def fill_scene(self, sc):
        vol = self.packed()
        sc.occ = vol.data_ptr()
        sc.occ_res[:] = [vol.shape[3], vol.shape[2], vol.shape[1]]
        cells = self.cell_or()
        sc.occ_cell = None if cells is None else cells.data_ptr()

    @_lib.device_guard
    def sample_alpha(self, norm_samples):
        _require_cuda(norm_samples, "sample_alpha")
        _last_dim(norm_samples, 7, "sample_alpha")
        c = _f32c(norm_samples)
        out = torch.empty(c.shape[:-1], device=c.device)
        sc = _lib.new_scene()
        self.fill_scene(sc)
        _call("ego_alpha_mask_sample", sc, c.data_ptr(), c.numel() // 7, out.data_ptr(), _lib.stream_handle())
        return out


class EgoNeRF(TensorBase):
    supports_need_alpha = True  # forward(..., need_alpha=False): see volume_renderer(keep_alpha=False)
    supports_marched_event = True  # forward(..., marched_event=ev): see renderer._render_to_host
    def __init__(self, aabb, gridSize, device, coordinates, **kargs):
        super().__init__(aabb, gridSize, device, coordinates, **kargs)
        assert isinstance(coordinates, YinYangSphericalCoords), "EgoNeRF needs YinYangSphericalCoords (EgoNeRF.py:522)"
        self.matMode_yin = self.matMode_yang = MAT_MODE
        self.vecMode_yin = self.vecMode_yang = VEC_MODE
        self.init_svd_volume(gridSize[0], device)
        if isinstance(self.renderModule, torch.nn.Module):
            self.renderModule._owner = _WeakOwner(self)
        self._scene_cache = None
        self._packed = None
        self._packed_versions = None
        self._sched_cache = {}
        self._mlp_precision = "f16f6"   # inference default; differentiable calls always use the three-term fp16 split
        # Differentiable calls of the tuned head keep the activations that only feed the weight-gradient products (x, h1, h2, dh1, dh2)
        # as halves (~2^-12 relative per operand against the reference's fp32 autograd, nothing above 65504).  True (or
        # EGO_TRAIN_FP32=1 in the environment at construction) trains it through the fp32 compatibility kernels instead: the parity
        # mode, several times slower.
        self.train_fp32_head = os.environ.get("EGO_TRAIN_FP32", "0") not in ("", "0")
        # Table gradients of a differentiable call: True (default) = sort the step's samples by texel cell and write every gradient texel
        # once, in a fixed order - no atomics, two runs return the same bits (csrc/ego_scatter_sorted.hip); False (or EGO_SCATTER=atomic
        # at construction) = the float-atomic scatters of rounds 1-4 (run-to-run differences of ~1e-6 of the largest gradient).
        self.deterministic_scatter = os.environ.get("EGO_SCATTER", "sorted") != "atomic"
        self._app_table_dtype = "f32"   # "f16": inference gathers appearance taps from a half-precision copy of the tables
        self._app16 = None              # (versions, [12 half tensors])
        # opt-in skipping (EgoNeRF.forward itself evaluates every sample; TensorBase.forward's semantics, tensorBase.py:464-487,
        # pinned to the reference by tests/golden/skip_semantics.npz; see include/egonerf_hip.h):
        self.use_alpha_mask = False          # apply self.alphaMask with TensorBase.forward's semantics (sigma = 0 where empty)
        self.early_termination_eps = 0.0     # > 0: zero the weight of samples behind transmittance < eps
        self.use_weight_thres = False        # TensorBase.forward's app skip: samples with weight <= rayMarch_weight_thres get rgb = 0
        # Exact skipping, on by default: 32-sample tiles whose weights are all exactly 0 are not shaded (ego_scene.weight_thres = 0).
        # The outputs are bit-identical - the reference adds w * rgb = 0 for such samples (EgoNeRF.py:583) - and behind an opaque
        # surface the fp32 transmittance underflows to 0 within a few samples, so this is what early termination amounts to.
        # EGO_EXACT_SKIP=0 in the environment switches it off for A/B measurements.
        self.skip_zero_weight_tiles = os.environ.get("EGO_EXACT_SKIP", "1") != "0"
        self.coarse_sigma_plane_yin, self.coarse_sigma_line_yin = [None] * 3, [None] * 3
        self.coarse_sigma_plane_yang, self.coarse_sigma_line_yang = [None] * 3, [None] * 3
        if self.coarse_sigma_grid_update_rule is not None:
            if self.coarse_sigma_grid_update_rule != "conv":
                raise NotImplementedError  # EgoNeRF.py:92-94
            if torch.device(device).type == "cuda":
                self.update_coarse_sigma_grid()

    # -- parameters -------------------------------------------------------------------------------------
    def init_one_svd(self, n_component, gridSize, scale, device):
        """EgoNeRF.py:102-122, channel-last memory."""
        shapes = []
        for _grid in ("yin", "yang"):
            for i in range(3):
                m0, m1 = MAT_MODE[i]
                shapes.append((n_component[i], gridSize[m1], gridSize[m0]))
            for i in range(3):
                shapes.append((n_component[i], gridSize[VEC_MODE[i]], 1))
        views = _carve_channel_last(shapes, device)   # [plane x3, line x3] yin, then yang: one buffer per field
        for v, (C_, H, W) in zip(views, shapes):
            v.copy_(scale * torch.randn(1, C_, H, W))
        params = [torch.nn.Parameter(v) for v in views]
        return [torch.nn.ParameterList(params[0:3]), torch.nn.ParameterList(params[3:6]),
                torch.nn.ParameterList(params[6:9]), torch.nn.ParameterList(params[9:12])]

    @torch.no_grad()
    def _reflatten_tables(self, kind: str):
        """Move the 12 tables of `kind` ("density" | "app") into one fresh buffer (after they were replaced one by one, e.g. by
        upsample_volume_grid): new Parameters, so an optimiser must be rebuilt afterwards (the reference does, train.py:380)."""
        lists = [getattr(self, f"{kind}_{what}_{g}") for g in ("yin", "yang") for what in ("plane", "line")]
        olds = [p for l in lists for p in l]
        views = _carve_channel_last([(p.shape[1], p.shape[2], p.shape[3]) for p in olds], olds[0].device)
        for v, p in zip(views, olds):
            v.copy_(p.data)
        k = 0
        for l in lists:
            for i in range(len(l)):
                l[i] = torch.nn.Parameter(views[k], requires_grad=olds[k].requires_grad)
                k += 1
        self._scene_cache = None

    # -- compact addressing guard -----------------------------------------------------------------------------
    def _table_lists(self, kind: str):
        return [getattr(self, f"{kind}_{what}_{g}") for g in ("yin", "yang") for what in ("plane", "line")]

    def _is_compact(self, kind: str) -> bool:
        """True iff the 12 tables of `kind` end within 4 GB of the lowest of them (what csrc/ego_device.h's 32-bit tap offsets
        need; ego_field_is_compact is the library's own check)."""
        ts = [p for l in self._table_lists(kind) for p in l]
        lo = min(t.data_ptr() for t in ts)
        hi = max(t.data_ptr() + t.numel() * t.element_size() for t in ts)
        return hi - lo < (1 << 32)

    @torch.no_grad()
    def _recompact_tables(self, kind: str):
        """Move the 12 tables of `kind` into one fresh buffer IN PLACE (`param.data = view`): the Parameter objects - and with
        them an optimiser's references and state - stay.  Needed after anything that re-allocates the parameters one by one
        (`Module._apply`: .to() / .cuda() / .float(); `load_state_dict(assign=True)`): separate hipMalloc segments of this size
        have no bounded distance, and the forward gathers address a tap as base + 32-bit offset."""
        olds = [p for l in self._table_lists(kind) for p in l]
        views = _carve_channel_last([(p.shape[1], p.shape[2], p.shape[3]) for p in olds], olds[0].device)
        for v, p in zip(views, olds):
            v.copy_(p.data)
            p.data = v
        self._scene_cache = None

    def _apply(self, fn, *args, **kwargs):
        """nn.Module._apply re-allocates every parameter on its own; restore what the kernels rely on afterwards: one buffer per
        field (compact addressing), the non-parameter device state (pooled density tables, packed weights, half copies) on
        the new device."""
        r = super()._apply(fn, *args, **kwargs)
        self._scene_cache = None
        self._packed = self._packed_versions = self._app16 = None
        self._sched_cache = {}
        if not hasattr(self, "density_plane_yin"):   # called from inside __init__ (renderModule.to(device)) before the tables exist
            return r
        dev = self.density_plane_yin[0].device
        self.device = dev
        if dev.type == "cuda":
            for kind in ("density", "app"):
                if not self._is_compact(kind):
                    self._recompact_tables(kind)
            stale = self.coarse_sigma_plane_yin[0] is None or self.coarse_sigma_plane_yin[0].device != dev
            if self.coarse_sigma_grid_update_rule == "conv" and stale:
                self.update_coarse_sigma_grid()
        return r

    def init_svd_volume(self, res, device):
        g = self.gridSize.tolist()
        (self.density_plane_yin, self.density_line_yin, self.density_plane_yang,
         self.density_line_yang) = self.init_one_svd(self.density_n_comp, g, 0.1, device)
        (self.app_plane_yin, self.app_line_yin, self.app_plane_yang,
         self.app_line_yang) = self.init_one_svd(self.app_n_comp, g, 0.1, device)
        self.basis_mat_yin = torch.nn.Linear(sum(self.app_n_comp), self.app_dim, bias=False).to(device)
        self.basis_mat_yang = torch.nn.Linear(sum(self.app_n_comp), self.app_dim, bias=False).to(device)

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001, lr_init_envmap=0.1):
        """EgoNeRF.py:139-156."""
        gv = []
        for g in ("yin", "yang"):
            gv += [{"params": getattr(self, f"density_line_{g}"), "lr": lr_init_spatialxyz},
                   {"params": getattr(self, f"density_plane_{g}"), "lr": lr_init_spatialxyz},
                   {"params": getattr(self, f"app_line_{g}"), "lr": lr_init_spatialxyz},
                   {"params": getattr(self, f"app_plane_{g}"), "lr": lr_init_spatialxyz},
                   {"params": getattr(self, f"basis_mat_{g}").parameters(), "lr": lr_init_network}]
        if isinstance(self.renderModule, torch.nn.Module):   # EgoNeRF.py:152 (RGBRender is a function)
            gv += [{"params": self.renderModule.parameters(), "lr": lr_init_network}]
        if self.envmap is not None:
            gv += [{"params": self.envmap.emission, "lr": lr_init_envmap}]
        return gv

    # -- regularisers (train.py:289-304) ----------------------------------------------------------------------
    def vectorDiffs(self, vector_comps):
        """EgoNeRF.py:189-196: sum over line tables of mean |off-diagonal of the component Gram matrix|."""
        from .losses import table_regulariser
        return table_regulariser("ortho", list(vector_comps), [1.0] * len(vector_comps))

    def vector_comp_diffs(self):
        """EgoNeRF.py:198-199."""
        from .losses import table_regulariser
        lines = (list(self.density_line_yin) + list(self.app_line_yin) + list(self.density_line_yang) + list(self.app_line_yang))
        return table_regulariser("ortho", lines, [1.0] * len(lines))

    def density_L1(self):
        """EgoNeRF.py:206-212."""
        from .losses import table_regulariser
        t = []
        for i in range(len(self.density_plane_yin)):
            t += [self.density_plane_yin[i], self.density_line_yin[i], self.density_plane_yang[i], self.density_line_yang[i]]
        return table_regulariser("l1", t, [1.0] * len(t))

    def _tv(self, reg, planes):
        from .losses import table_regulariser
        w = float(getattr(reg, "TVLoss_weight", 1.0))
        return table_regulariser("tv", planes, [1e-2 * w] * len(planes))

    def TV_loss_density(self, reg):
        """EgoNeRF.py:214-220: planes only, 1e-2 each; `reg` is a TVLoss (its weight is honoured, its forward is fused here)."""
        return self._tv(reg, [p for i in range(3) for p in (self.density_plane_yin[i], self.density_plane_yang[i])])

    def TV_loss_app(self, reg):
        """EgoNeRF.py:222-228."""
        return self._tv(reg, [p for i in range(3) for p in (self.app_plane_yin[i], self.app_plane_yang[i])])

    # -- coarse-to-fine upsampling (train.py:371-385) -------------------------------------------------------------
    @torch.no_grad()
    def up_sampling_VM(self, plane_coef, line_coef, res_target):
        """EgoNeRF.py:415-426."""
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            plane_coef[i] = self.coordinates.up_sampling_VM(plane_coef[i].data, res_target=res_target, ids=[m1, m0])
            line_coef[i] = self.coordinates.up_sampling_VM(line_coef[i].data, res_target=res_target, ids=[VEC_MODE[i]])
        return plane_coef, line_coef

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        """EgoNeRF.py:428-435.  The caller then calls coordinates.set_resolution(res_target) (train.py:376-377; note that
        call resets r0 to 0.05 unless r0 is passed) and rebuilds the optimiser."""
        self.app_plane_yin, self.app_line_yin = self.up_sampling_VM(self.app_plane_yin, self.app_line_yin, res_target)
        self.density_plane_yin, self.density_line_yin = self.up_sampling_VM(self.density_plane_yin, self.density_line_yin, res_target)
        self.app_plane_yang, self.app_line_yang = self.up_sampling_VM(self.app_plane_yang, self.app_line_yang, res_target)
        self.density_plane_yang, self.density_line_yang = self.up_sampling_VM(self.density_plane_yang, self.density_line_yang, res_target)
        self._reflatten_tables("app")
        self._reflatten_tables("density")
        self.update_stepSize(res_target)
        self._scene_cache = None
        print(f"upsamping to {res_target}")

    @torch.no_grad()
    @_lib.device_guard
    def update_coarse_sigma_grid(self):
        """2x average-pooled density tables (EgoNeRF.py:124-133), kept channel-last."""
        if self.coarse_sigma_grid_update_rule != "conv":
            raise NotImplementedError
        st = _lib.stream_handle()
        srcs = [(g, what, i, getattr(self, f"density_{what}_{g}")[i]) for g in ("yin", "yang") for what in ("plane", "line") for i in range(3)]
        for *_k, src in srcs:
            _require_cuda(src, "update_coarse_sigma_grid")
        shapes = [(src.shape[1], src.shape[2] // 2, 1 if src.shape[3] == 1 else src.shape[3] // 2) for *_k, src in srcs]
        # The pooled tables are refreshed IN PLACE while their shapes stand (every training step, train.py:356-357): no allocation
        # per step, the scene struct stays valid, and a captured hipGraph of the iteration reads on replay k + 1 what replay k wrote
        # (freshly allocated tables would be invisible to the already captured forward).
        cur = [getattr(self, f"coarse_sigma_{what}_{g}")[i] for (g, what, i, _src) in srcs]
        same = all(c is not None and c.device == src.device and tuple(c.shape) == (1, C_, H, W)
                   for c, (C_, H, W), (*_k, src) in zip(cur, shapes, srcs))
        if same:
            # ... and only while they still form ONE carved buffer: a caller may have assigned coarse_sigma_* one by one (separate
            # allocations have no bounded distance, and the march addresses a tap as base + 32-bit offset: ADVICE r03)
            lo = min(c.data_ptr() for c in cur)
            hi = max(c.data_ptr() + c.numel() * c.element_size() for c in cur)
            same = hi - lo < (1 << 32) and all(c.dtype == torch.float32 and c.permute(0, 2, 3, 1).is_contiguous() for c in cur)
        if not same:
            dsts = _carve_channel_last(shapes, srcs[0][3].device)   # one buffer: compact addressing like the full tables
            for (g, what, i, src), dst in zip(srcs, dsts):
                getattr(self, f"coarse_sigma_{what}_{g}")[i] = dst
            self._scene_cache = None
        # all 12 tables in one launch (this runs after every training step, train.py:356-357)
        fs, fd = _lib.VmField(), _lib.VmField()
        res = [int(v) for v in self.gridSize.tolist()]
        self._fill_field(fs, "density", self.density_n_comp, res)
        self._fill_field(fd, "density", self.density_n_comp, [r // 2 for r in res], coarse=True)
        _call("ego_avgpool_field", C.byref(fs), C.byref(fd), st)

    @property
    def is_tuned_shape(self) -> bool:
        """True for the model shape every shipped config resolves to (head_is_tuned AND 16 density components): the MFMA / team-gather
        kernels end to end, forward and backward.  Any other shape opt.py:87-100 can produce (n_lamb_sigma / n_lamb_sh multiples of 4
        up to 48, data_dim_color <= 32, featureC 64 | 128, view_pe / fea_pe <= 8, shadingMode 'MLP' / 'RGB') renders and trains through
        the fp32 compatibility kernels for the part that differs: same results to fp32 rounding, 13-32 x slower (tools/generic_timing.py:
        8.8-21 ms against 0.67 ms per 4096 x 512 inference step, 89-167 ms against 6.4 ms per 8192-ray training step)."""
        return self.head_is_tuned and self.density_n_comp[0] == 16

    @property
    def head_is_tuned(self) -> bool:
        """The appearance head alone (= the C library's ego_shape_is_tuned): MLP_Fea 150 -> 128 -> 128 -> 3 on 48 components / app_dim 27
        with view_pe = fea_pe = 2 shades, and trains, through the MFMA kernels whatever the density field's component count (16 takes the
        tuned march / density scatter, any other multiple of 4 the compatibility march around the same shade kernels)."""
        return (self.shadingMode, self.app_dim, self.app_n_comp[0], self.featureC, self.view_pe, self.fea_pe) == ("MLP_Fea", 27, 48, 128, 2, 2)

    @property
    def mlp_precision(self) -> str:
        """Arithmetic of the basis/MLP products:
        "f16f6" (default): layers 1 and 2 with the main term in fp16 and the two correction terms on the fp6 (e2m3) path of the
                 block-scaled MFMA, per-lane block scales taken from the activations at run time; composited max |d RGB| ~1.7e-5
                 (bar: 1e-4); inference only — a differentiable call uses "f16x3";
        "f16f8": the same split with both correction terms in one block-scaled fp8 (e4m3, fixed scales) MFMA per pair of k-steps
                 (round 2's default; 3-4 % slower shade kernel, saturates above 448);
        "f16x3": three fp16 MFMAs per product, fp32-grade (2e-7);
        "f32":   fp32-input MFMA, bit-for-bit fp32 FMA chains (2.8x slower)."""
        return self._mlp_precision

    @mlp_precision.setter
    def mlp_precision(self, value: str):
        if value not in ("f16x3", "f16f8", "f16f6", "f32"):
            raise ValueError("mlp_precision must be 'f16x3', 'f16f8', 'f16f6' or 'f32'")
        self._mlp_precision = value
        self._scene_cache = None

    @torch.no_grad()
    def check_mlp_precision(self, rays: torch.Tensor, **forward_kw) -> dict:
        """Renders `rays` with the current arithmetic and with the fp32-grade "f16x3" one and returns the largest colour
        difference.  "f16f8" is 2^-16-relative per product: its error scales with the magnitudes inside the MLP (1.4e-5 composited
        on the bench scene, 2e-5 per sample at nn.Linear-default x 3 weights, 3e-3 per sample when the logits reach +-50), so a
        trained checkpoint should be checked once on a few thousand rays and switched to "f16x3" if this exceeds its budget."""
        keep = self._mlp_precision
        try:
            got = self.forward(rays, **forward_kw)[0]
            self.mlp_precision = "f16x3"
            ref = self.forward(rays, **forward_kw)[0]
        finally:
            self.mlp_precision = keep
        d = (got - ref).abs()
        return dict(mlp_precision=keep, max_abs_rgb_diff_vs_f16x3=float(d.max()) if d.numel() else 0.0,
                    mean_abs_rgb_diff_vs_f16x3=float(d.mean()) if d.numel() else 0.0, rays=int(rays.shape[0]))

    @property
    def app_table_dtype(self) -> str:
        """Storage of the appearance tables the inference gather reads: "f32" (the parameters themselves) or "f16" (a
        half-precision shadow copy, refreshed when the parameters change; interpolation stays fp32).  Training always
        uses the fp32 tables."""
        return self._app_table_dtype

    @app_table_dtype.setter
    def app_table_dtype(self, value: str):
        if value not in ("f32", "f16"):
            raise ValueError("app_table_dtype must be 'f32' or 'f16'")
        self._app_table_dtype = value
        self._scene_cache = None

    def _app_tables(self) -> List[torch.Tensor]:
        out = []
        for g in ("yin", "yang"):
            out += list(getattr(self, f"app_plane_{g}")) + list(getattr(self, f"app_line_{g}"))
        return out

    def _fill_app16(self, sc):
        tabs = self._app_tables()
        ver = tuple((t.data_ptr(), t._version) for t in tabs)
        if self._app16 is None or self._app16[0] != ver:
            # [1,H,W,C] channel-last memory of the (1,C,H,W) parameters, converted to half (a cast, done when weights change), in
            # one buffer
            halves = _carve_channel_last([(t.shape[1], t.shape[2], t.shape[3]) for t in tabs], tabs[0].device, torch.float16)
            for hv, t in zip(halves, tabs):
                hv.copy_(t.detach())
            self._app16 = (ver, halves)
        halves = self._app16[1]
        sc.app16.n_comp = self.app_n_comp[0]
        sc.app16.res[:] = self.gridSize.tolist()
        for gi in range(2):
            for i in range(3):
                sc.app16.plane[gi][i] = halves[gi * 6 + i].data_ptr()
                sc.app16.line[gi][i] = halves[gi * 6 + 3 + i].data_ptr()
        sc.app_f16 = 1

    # -- C-ABI scene ----------------------------------------------------------------------------------------
    def _mlp_tensors(self) -> List[torch.Tensor]:
        if self.shadingMode == "RGB":   # no network: the packed blob holds the two basis matrices only
            return [self.basis_mat_yin.weight, self.basis_mat_yang.weight]
        m = self.renderModule.mlp
        return [m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias, self.basis_mat_yin.weight,
                self.basis_mat_yang.weight]

    def _fill_field(self, f: "_lib.VmField", prefix: str, n_comp, res, coarse=False):
        if len(set(n_comp)) != 1:
            raise NotImplementedError("per-plane component counts must be equal (true for every shipped config)")
        f.n_comp = n_comp[0]
        f.res[:] = res
        for gi, g in enumerate(("yin", "yang")):
            planes = getattr(self, f"coarse_sigma_plane_{g}" if coarse else f"{prefix}_plane_{g}")
            lines = getattr(self, f"coarse_sigma_line_{g}" if coarse else f"{prefix}_line_{g}")
            for i in range(3):
                if planes[i] is None:
                    raise RuntimeError("coarse density tables missing: call update_coarse_sigma_grid()")
                f.plane[gi][i] = _table_ptr(planes[i])
                f.line[gi][i] = _table_ptr(lines[i])

    @_lib.device_guard
    def scene(self, training: bool = False) -> "_lib.Scene":
        """The ego_scene struct for the current parameters (re-packs the MFMA weights when they changed).

        training=True is what the differentiable path asks for (train.RenderFunction, forward and backward): the fp32 tables, every
        sample shaded like EgoNeRF.forward (the appearance skip is an inference option), all three fp16 terms of every product
        (mlp_precision "f16x3"), and only the regions of the packed blob that arithmetic reads (ego_pack_mlp_for).

        The struct holds raw device pointers: to the parameter tables (optimiser steps write them in place), and to the pooled density
        tables, which `update_coarse_sigma_grid()` also refreshes IN PLACE while their shapes stand - so a struct returned earlier, or a
        captured hipGraph that embeds it, reads the refreshed values (what GraphedTrainStep relies on).  The packed MLP / basis images
        are different: a SNAPSHOT taken by this call, one blob per mode (training / inference) - a struct returned earlier keeps reading
        the weights as of its own pack (a captured graph that contains the pack kernel re-packs on every replay); call scene() again
        after the weights changed.  A holder that needs a frozen
        snapshot must copy the tables; re-allocation (another shape / device, tables assigned one by one) invalidates the cached
        struct and the next scene() call builds a new one."""
        dev = self.density_plane_yin[0].device
        if dev.type != "cuda":
            raise RuntimeError(f"model parameters are on {dev}; the EgoNeRF hot path runs only on the HIP device")
        for kind in ("density", "app"):  # e.g. after load_state_dict(assign=True); .to() / .cuda() / .float() are handled in _apply
            if not self._is_compact(kind):
                self._recompact_tables(kind)
        mlp = self._mlp_tensors()
        versions = tuple((t.data_ptr(), t._version) for t in mlp)
        if self._app_table_dtype == "f16":  # the half copy follows the appearance tables' versions
            versions += tuple((t.data_ptr(), t._version) for t in self._app_tables())
        co = self.coordinates
        luts = [co.lut_device(dev, 2)] + ([] if co.interval_th else [co.lut_device(dev, None)])  # held by the cache entry below
        keys = tuple(p.data_ptr() for p in self.parameters()) + (None if self.envmap is None else self.envmap.emission.data_ptr(),
                                                                  self.use_alpha_mask, id(self.alphaMask), float(self.early_termination_eps),
                                                                  float(self.rayMarch_weight_thres) if self.use_weight_thres else None,
                                                                  bool(self.skip_zero_weight_tiles), co.N_r, float(co.r0), float(co.far[0]), tuple(co.center.tolist()),
                                                                  tuple(t.data_ptr() for t in luts), float(self.distance_scale),
                                                                  float(self.density_shift), self.fea2denseAct, bool(training))
        if self._scene_cache is not None and self._scene_cache[0] == keys and self._packed_versions == versions:
            return self._scene_cache[1]
        lib = _lib.load()
        sc = _lib.new_scene()
        co.fill_scene(sc, dev)
        sc.act_softplus = int(self.fea2denseAct == "softplus")
        sc.density_shift, sc.distance_scale = float(self.density_shift), float(self.distance_scale)
        g = self.gridSize.tolist()
        self._fill_field(sc.density, "density", self.density_n_comp, g)
        self._fill_field(sc.app, "app", self.app_n_comp, g)
        if self.coarse_sigma_plane_yin[0] is not None:
            self._fill_field(sc.density_coarse, "density", self.density_n_comp, [v // 2 for v in g], coarse=True)
        holds = [t.detach().contiguous() for t in mlp]
        if self.shadingMode == "RGB":
            sc.head = 1   # EGO_HEAD_RGB: mlp_w / mlp_b stay null, mlp_in = mlp_hidden = 0
        else:
            sc.mlp_w[:] = [holds[0].data_ptr(), holds[2].data_ptr(), holds[4].data_ptr()]
            sc.mlp_b[:] = [holds[1].data_ptr(), holds[3].data_ptr(), holds[5].data_ptr()]
        sc.basis[:] = [holds[-2].data_ptr(), holds[-1].data_ptr()]
        sc.app_dim = self.app_dim
        sc.mlp_in, sc.mlp_hidden = self.head_in_mlpC, self.head_hidden
        sc.view_pe, sc.fea_pe = (self.view_pe, self.head_fea_pe) if self.shadingMode != "RGB" else (0, 0)
        sc.mlp_precision = 0 if training else {"f16x3": 0, "f32": 1, "f16f8": 2, "f16f6": 3}[self._mlp_precision]
        # packed weights: the MFMA fragment layouts for the tuned shape (27 / 48 / 150 / 128 / 2 / 2, every shipped config), the fp32
        # layout of the any-shape compatibility kernels otherwise (csrc/ego_generic.hip; the library reports the size for the shape)
        sc.app.n_comp = self.app_n_comp[0]
        need = lib.ego_packed_floats_scene(C.byref(sc))
        if need <= 0:
            raise RuntimeError("ego_packed_floats_scene rejected the scene")
        # one blob per mode: a training pack writes only the regions the f16x3 training kernels read, so it must never land
        # in the blob an earlier INFERENCE struct (or a captured preview-render graph) points to - that holder would mix fresh fp16 main
        # terms with stale fp8 / fp6 correction terms
        if self._packed is None:
            self._packed = {}
        blob = self._packed.get(bool(training))
        if blob is None or blob.device != dev or blob.numel() != need:
            blob = self._packed[bool(training)] = torch.zeros(need, device=dev)
        _call("ego_pack_mlp_for", sc, blob.data_ptr(), int(training), _lib.stream_handle())
        sc.packed = blob.data_ptr()
        if self._app_table_dtype == "f16" and not training:
            self._fill_app16(sc)
        if self.use_alpha_mask and self.alphaMask is not None:
            self.alphaMask.fill_scene(sc)
        sc.term_eps = float(self.early_termination_eps)
        sc.weight_thres = float(self.rayMarch_weight_thres) if self.use_weight_thres else (0.0 if self.skip_zero_weight_tiles else -1.0)
        if training:
            sc.weight_thres = -1.0
        if self.envmap is not None:
            em = self.envmap.emission.detach()
            if not em.is_contiguous():
                raise RuntimeError("envmap.emission must be contiguous [3][2h][h]")
            sc.envmap, sc.envmap_h = em.data_ptr(), em.shape[2]
        self._scene_cache = (keys, sc, holds, luts)
        self._packed_versions = versions
        return sc

    # -- stage methods (reference public API) ----------------------------------------------------------------
    def _sched(self, n_samples: int, device) -> torch.Tensor:
        near, far = self.near_far
        key = (n_samples, str(device), float(self.coordinates.r0), float(near), float(far),
               bool(self.coordinates.interval_th))  # set_resolution() may move r0 (coordinates.py:214)
        if key not in self._sched_cache:
            self._sched_cache[key] = self.coordinates.sample_schedule(near, far, n_samples).to(device)
        return self._sched_cache[key]

    def sample_ray_z(self, rays: torch.Tensor, n_samples: int, jitter: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Distances of TensorBase.sample_ray (tensorBase.py:308-327): aabb entry clamped to [near, far] + stepSize * (k [+ U])."""
        o, d = rays[:, :3], rays[:, 3:6]
        near, far = self.near_far
        aabb = self.aabb.to(rays.device, torch.float32)
        vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
        t_min = torch.minimum((aabb[1] - o) / vec, (aabb[0] - o) / vec).amax(-1).clamp(min=near, max=far)
        rng = torch.arange(n_samples, device=rays.device)[None].float()
        if jitter is not None:
            rng = rng.repeat(rays.shape[0], 1) + jitter.to(rays.device, torch.float32)
        step = self.stepSize.to(rays.device, torch.float32) if torch.is_tensor(self.stepSize) else float(self.stepSize)
        z = t_min[:, None] + step * rng
        return z.expand(rays.shape[0], n_samples).contiguous()

    def sample_ray(self, rays_o, rays_d, is_train=True, N_samples=-1, jitter: Optional[torch.Tensor] = None):
        """tensorBase.py:308-327 -> (rays_pts [N,S,3], interpx [N,S], inside-aabb mask [N,S])."""
        N_samples = N_samples if N_samples > 0 else self.nSamples
        if is_train and jitter is None:
            jitter = torch.rand(rays_o.shape[0], N_samples)
        z = self.sample_ray_z(torch.cat([rays_o, rays_d], -1).float(), N_samples, jitter if is_train else None)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., None]
        aabb = self.aabb.to(pts.device)
        return pts, z, ~((aabb[0] > pts) | (pts > aabb[1])).any(dim=-1)

    def _plain_exp_train_z(self, n: int, jitter: torch.Tensor) -> torch.Tensor:
        """Plain exponential schedule in training (EgoNeRF.py:59-67): the noise goes into the exponent and the distances are an
        exclusive prefix sum; the reference's own expression, handed to the kernels as explicit distances [N, n]."""
        near, far = float(self.near_far[0]), float(self.near_far[1])
        ratio = 1 + (math.pi / 2.0) / n
        r0 = (far - near) * (ratio - 1) / (pow(ratio, n) - 1)
        rng = torch.arange(n, device=jitter.device)[None].float() + _f32c(jitter)
        tri = torch.tril(torch.ones(n, n, device=jitter.device), diagonal=-1).T
        return (near + torch.pow(ratio, rng) @ tri * r0).contiguous()

    @_lib.device_guard
    def sample_ray_exp(self, rays_o, rays_d, is_train=True, N_samples=-1, jitter: Optional[torch.Tensor] = None):
        """EgoNeRF.py:56-87 -> (rays_pts [N,S,3], interpx [N,S], ~mask_outbbox [N,S]).  `jitter` [N,S] pins the
        training noise (the reference draws it with torch.rand_like on the CPU generator)."""
        _require_cuda(rays_o, "sample_ray_exp")
        self.coordinates._require_supported()
        N, S = rays_d.shape[-2], N_samples
        rays = torch.cat([_f32c(rays_o), _f32c(rays_d)], -1).contiguous()
        if is_train and jitter is None:
            jitter = torch.rand(N, S).to(rays.device)
        jit = _f32c(jitter) if is_train else None
        if is_train and not self.coordinates.interval_th:
            z = self._plain_exp_train_z(S, jit.to(rays.device))
            xyz = rays[:, None, :3] + rays[:, None, 3:6] * z[..., None]
        else:
            xyz = torch.empty(N, S, 3, device=rays.device)
            z = torch.empty(N, S, device=rays.device)
            _call("ego_sample_ray_exp", rays.data_ptr(), self._sched(S, rays.device).data_ptr(), _lib.ptr(jit),
                  float(self.near_far[0]), N, S, xyz.data_ptr(), z.data_ptr(), _lib.stream_handle())
        aabb = self.aabb.to(rays.device)
        mask_outbbox = ((aabb[0] > xyz) | (xyz > aabb[1])).any(dim=-1)
        return xyz, z, ~mask_outbbox

    @_lib.device_guard
    def _density(self, coords_sampled, coarse: int):
        _require_cuda(coords_sampled, "compute_densityfeature")
        _last_dim(coords_sampled, 7, "compute_densityfeature")
        c = _f32c(coords_sampled)
        out = torch.empty(c.shape[:-1], device=c.device)
        _call("ego_density_feature", self.scene(), c.data_ptr(), c.numel() // 7, coarse, out.data_ptr(), _lib.stream_handle())
        return out

    def compute_densityfeature(self, coords_sampled):
        """EgoNeRF.py:291-347: [...,7] normalised coords -> [...]."""
        return self._density(coords_sampled, 0)

    def compute_coarse_densityfeature(self, coords_sampled, coarse_sigma_grid_update_rule="conv"):
        """EgoNeRF.py:232-289."""
        return self._density(coords_sampled, 1)

    @_lib.device_guard
    def compute_appfeature(self, coords_sampled):
        """EgoNeRF.py:349-413: [...,7] -> [..., app_dim]."""
        _require_cuda(coords_sampled, "compute_appfeature")
        _last_dim(coords_sampled, 7, "compute_appfeature")
        c = _f32c(coords_sampled)
        out = torch.empty(*c.shape[:-1], self.app_dim, device=c.device)
        _call("ego_app_feature", self.scene(), c.data_ptr(), c.numel() // 7, out.data_ptr(), _lib.stream_handle())
        return out

    # -- occupancy ------------------------------------------------------------------------------
    @torch.no_grad()
    def compute_alpha(self, norm_locs, length=1):
        """tensorBase.py:421-436: alpha on arbitrary normalised locations, skipping what an existing mask marks empty."""
        sigma = torch.zeros(norm_locs.shape[:-1], device=norm_locs.device)
        if self.alphaMask is not None:
            keep = self.alphaMask.sample_alpha(norm_locs) > 0
        else:
            keep = torch.ones_like(sigma, dtype=torch.bool)
        if bool(keep.any()):
            sigma[keep] = self.feature2density(self.compute_densityfeature(norm_locs[keep]))
        return 1 - torch.exp(-sigma * length)

    @torch.no_grad()
    def getDenseAlpha(self, gridSize=None):
        """EgoNeRF.py:437-465: alpha on the grid lattice of both grids, step length = stepSize (no distance_scale)."""
        g = self.gridSize.tolist() if gridSize is None else list(gridSize)
        dev = self.density_plane_yin[0].device
        lin = [torch.linspace(0, 1, n) for n in g]
        norm = (torch.stack(torch.meshgrid(*lin, indexing="ij"), -1) * 2 - 1).to(dev)
        zeros3, flag = torch.zeros_like(norm), torch.zeros_like(norm[..., :1])
        yin = torch.cat([norm, zeros3, flag], -1).view(-1, 7)
        yang = torch.cat([zeros3, norm, flag + 1], -1).view(-1, 7)
        step = float(self.stepSize)
        return self.compute_alpha(yin, step).view(g), self.compute_alpha(yang, step).view(g)

    @torch.no_grad()
    def updateAlphaMask(self, gridSize=None):
        """EgoNeRF.py:467-489: clamp, 3x3x3 max-pool, threshold at alphaMask_thres -> YinYangAlphaGridMask.  Building the
        mask does not switch it on: set `use_alpha_mask = True` to apply it."""
        g = self.gridSize.tolist() if gridSize is None else list(gridSize)
        vols = []
        for a in self.getDenseAlpha(g):
            a = a.clamp(0, 1).transpose(0, 2).contiguous()[None, None]
            a = torch.nn.functional.max_pool3d(a, kernel_size=3, padding=1, stride=1).view(g[::-1])
            vols.append((a >= self.alphaMask_thres).float())
        self.alphaMask = YinYangAlphaGridMask(self.device, vols[0], vols[1])
        self._scene_cache = None
        return float((vols[0].sum() + vols[1].sum()) / (2 * g[0] * g[1] * g[2]))

    # -- the hot path -------------------------------------------------------------------------------------------
    @_lib.device_guard
    def forward(self, rays_chunk, white_bg=True, is_train=False, ndc_ray=False, n_coarse=-1, n_fine=0, exp_sampling=False,
                pretrain_envmap=False, pivotal_sample_th=0.0, resampling=False, use_coarse_sample=True, interval_th=False,
                jitter: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None, need_alpha: bool = True,
                marched_event: Optional["torch.cuda.Event"] = None):
        out = self._forward(rays_chunk, is_train, ndc_ray, n_coarse, n_fine, exp_sampling, pretrain_envmap, resampling, use_coarse_sample,
                            jitter, u, need_alpha, marched_event)
        if marched_event is not None and not getattr(self, "_marched_recorded", False):
            marched_event.record()   # a path without a march of its own (training, envmap pre-training, ...): "marched" = done
        self._marched_recorded = False
        return out

    def _forward(self, rays_chunk, is_train, ndc_ray, n_coarse, n_fine, exp_sampling, pretrain_envmap, resampling, use_coarse_sample,
                 jitter, u, need_alpha, marched_event):
        """EgoNeRF.forward (EgoNeRF.py:491-602) -> (rgb_map [N,3], depth_map [N], bg_map|None, env_map|None,
        alpha [N, S(+1)]).  `white_bg`, `pivotal_sample_th`, `interval_th` are accepted and unused, as in the
        reference.  `jitter` [N,n_coarse] / `u` [N,n_fine] pin the is_train noise.  `need_alpha=False` (eval only; what
        `volume_renderer(keep_alpha=False)` passes) returns None for the per-sample alpha, which also lets the ray march stop
        evaluating a ray once its transmittance is exactly 0 (the remaining weights are exactly 0 either way).  `marched_event` (a created
        torch.cuda.Event; volume_renderer's host hand-over passes it): recorded on the current stream behind the call's last march, ahead
        of its shade kernel (ego_render_args.marched) - or at the end of a call that takes another path."""
        _require_cuda(rays_chunk, "EgoNeRF.forward")
        if rays_chunk.dim() != 2 or rays_chunk.shape[1] < 6:
            raise IndexError(f"EgoNeRF.forward: rays_chunk must be [N, >=6] (origin, direction), got {tuple(rays_chunk.shape)}")
        if pretrain_envmap:
            return self.envmap.get_radiance(rays_chunk[:, 3:6])
        if ndc_ray:
            raise NotImplementedError  # EgoNeRF.py:503-504
        rays = _f32c(rays_chunk[:, :6])
        N, dev = rays.shape[0], rays.device
        z_coarse = None
        if not exp_sampling:
            # TensorBase.sample_ray (tensorBase.py:308-327): the per-ray uniform schedule is computed here and handed over as
            # explicit distances; the noise (`rng += rand`) is `jitter`
            if is_train and jitter is None:
                jitter = torch.rand(N, n_coarse)  # CPU generator like tensorBase.py:320
            z_coarse = self.sample_ray_z(rays, n_coarse, jitter.to(dev) if is_train else None)
            if not is_train:
                if N and not bool((z_coarse[:, 0] == z_coarse[0, 0]).all()):
                    # rays that enter the aabb at different distances: the reference places the first-pass samples per ray but
                    # measures every ray with ray 0's distances (EgoNeRF.py:515-516) - evaluated the same way, stage by stage
                    return self._forward_eval_ray0_distances(rays, z_coarse, int(n_coarse), int(n_fine), bool(resampling),
                                                             bool(use_coarse_sample), need_alpha)
            jitter = None
        if is_train and exp_sampling and not self.coordinates.interval_th:
            if jitter is None:
                jitter = torch.rand(N, n_coarse)
            z_coarse = self._plain_exp_train_z(n_coarse, jitter.to(dev))
            jitter = None
        if is_train:
            if jitter is None:
                jitter = torch.rand(N, n_coarse).to(dev)  # CPU generator like EgoNeRF.py:81
            if resampling and u is None:
                u = torch.rand(N, n_fine, device=dev)  # device generator like ray_utils.py:169
            jitter = _f32c(jitter)
            u = _f32c(u) if resampling else None
        else:
            jitter = u = None
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # any supported shape trains: the tuned one through the MFMA kernels, the others through the fp32 compatibility kernels
            from .train import render_train  # differentiable path: keeps activations, backward in HIP (egonerf_amd/train.py)
            return render_train(self, rays, n_coarse, n_fine, resampling, use_coarse_sample, jitter, u, z_coarse)
        sc = self.scene()
        args = _lib.RenderArgs()
        args.n_coarse, args.n_fine = int(n_coarse), int(n_fine)
        args.resampling, args.use_coarse_sample = int(bool(resampling)), int(bool(use_coarse_sample))
        args.r_sched = self._sched(n_coarse, dev).data_ptr()
        args.near_ = float(self.near_far[0])
        if z_coarse is not None:
            args.z_coarse = z_coarse.data_ptr()
        if jitter is not None:
            args.jitter = jitter.data_ptr()
        if u is not None:
            args.u = u.data_ptr()
        if marched_event is not None and N:
            args.marched = marched_event.cuda_event
            self._marched_recorded = True
        S = (n_coarse + n_fine if use_coarse_sample else n_fine) if resampling else n_coarse
        lib = _lib.load()
        ws_bytes = lib.ego_render_workspace_bytes(N, C.byref(args))
        if ws_bytes < 0:
            raise RuntimeError("ego_render_workspace_bytes rejected the arguments (n_coarse < 2?)")
        ws = torch.empty(max(ws_bytes, 4) // 4, device=dev)
        has_env = self.envmap is not None
        rgb_map = torch.empty(N, 3, device=dev)
        depth = torch.empty(N, device=dev)
        alpha = torch.empty(N, S + int(has_env), device=dev) if need_alpha else None
        bg_map = torch.empty(N, 3, device=dev) if has_env else None
        env_map = torch.empty(N, 3, device=dev) if has_env else None
        _call("ego_render_forward", sc, C.byref(args), rays.data_ptr(), N, ws.data_ptr(), rgb_map.data_ptr(), depth.data_ptr(),
              _lib.ptr(alpha), _lib.ptr(bg_map), _lib.ptr(env_map), _lib.stream_handle())
        return rgb_map, depth, bg_map, env_map, alpha

    def _forward_eval_ray0_distances(self, rays, z_pos, n_coarse, n_fine, resampling, use_coarse_sample, need_alpha):
        """Eval with exp_sampling=False and rays whose aabb entry distances differ (rays starting outside the box).  The reference
        (EgoNeRF.py:506-518) samples every ray at its OWN distances `t_min_i + k * stepSize` but then overwrites the distances of all
        rays with ray 0's: the first-pass intervals, the depth integral and - with resampling - the proposal bins and therefore the
        fine sample positions `o + d * z_fine` all come from ray 0's schedule.  Reproduced here with the stage entry points: a
        first-pass sample of ray i sits at o_i + d_i * (z_0[k] + (t_min_i - t_min_0)), i.e. on the ray with the origin moved by
        d_i * (t_min_i - t_min_0) and measured with z_0; the fine pass runs on the ORIGINAL origins with the merged distances."""
        lib, st, sc = _lib.load(), _lib.stream_handle(), self.scene()
        N, dev = rays.shape[0], rays.device
        f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        z0 = z_pos[0:1].expand(N, n_coarse).contiguous()
        shifted = rays.clone()
        shifted[:, :3] = rays[:, :3] + rays[:, 3:6] * (z_pos[:, :1] - z_pos[0:1, :1])
        S = (n_coarse + n_fine if use_coarse_sample else n_fine) if resampling else n_coarse
        has_env = self.envmap is not None
        astride = S + int(has_env)
        alpha = f(N, astride) if need_alpha else None
        w, bg, crd, rgb = f(N, S), f(N), f(N, S, 4), f(N, S, 3)
        use_flags = bool(sc.occ) or sc.term_eps > 0 or sc.weight_thres >= 0
        act = torch.zeros(N * S // 32 + 1, device=dev, dtype=torch.uint8) if use_flags else None
        near = float(self.near_far[0])
        if resampling:
            wc, z = f(N, n_coarse), f(N, S)
            _call("ego_march_density", sc, shifted.data_ptr(), N, n_coarse, z0.data_ptr(), None, None, near, 1, None, None, 0, wc.data_ptr(),
                  None, None, None, None, st)
            _call("ego_sample_pdf_merge", z0.data_ptr(), wc.data_ptr(), None, N, n_coarse, n_fine, int(use_coarse_sample), z.data_ptr(), None, st)
            _call("ego_march_density", sc, rays.data_ptr(), N, S, z.data_ptr(), None, None, near, 2, None, _lib.ptr(alpha), astride, w.data_ptr(),
                  bg.data_ptr(), crd.data_ptr(), None, _lib.ptr(act), st)
        else:
            z = z0
            _call("ego_march_density", sc, shifted.data_ptr(), N, S, z0.data_ptr(), None, None, near, 0, None, _lib.ptr(alpha), astride, w.data_ptr(),
                  bg.data_ptr(), crd.data_ptr(), None, _lib.ptr(act), st)
        _call("ego_shade", sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N, S, rgb.data_ptr(), None, _lib.ptr(act), st)
        rgb_map, depth = f(N, 3), f(N)
        bg_map, env_map = (f(N, 3), f(N, 3)) if has_env else (None, None)
        _call("ego_composite", sc, rays.data_ptr(), z.data_ptr(), w.data_ptr(), bg.data_ptr(), rgb.data_ptr(), N, S, rgb_map.data_ptr(),
              depth.data_ptr(), _lib.ptr(bg_map), _lib.ptr(env_map), None, st)
        return rgb_map, depth, bg_map, env_map, alpha

    # -- checkpoints (EgoNeRF.py:158-187) -----------------------------------------------------------------------------
    def save(self, path, global_step):
        ckpt = {"kwargs": self.get_kwargs(), "state_dict": {k: v.contiguous() for k, v in self.state_dict().items()},
                "global_step": global_step}
        if self.alphaMask is not None:  # packed bit volumes, EgoNeRF.py:161-167
            for g in ("yin", "yang"):
                vol = getattr(self.alphaMask, f"alpha_volume_{g}").bool().cpu().numpy()
                ckpt.update({f"alphaMask_{g}.shape": vol.shape, f"alphaMask_{g}.mask": np.packbits(vol.reshape(-1))})
        if self.envmap is not None:
            ckpt.update({"envmap.emission": self.envmap.emission.detach().cpu().numpy(),
                         "envmap_res_H": self.envmap.emission.shape[2]})
        from .compat import reference_pickle_paths
        with reference_pickle_paths():  # kwargs' coordinates / envmap objects go out under the reference's class paths
            torch.save(ckpt, path)

    def load(self, ckpt):
        if "alphaMask_yin.shape" in ckpt:  # EgoNeRF.py:175-180
            vols = []
            for g in ("yin", "yang"):
                shape = tuple(ckpt[f"alphaMask_{g}.shape"])
                bits = np.unpackbits(ckpt[f"alphaMask_{g}.mask"])[: int(np.prod(shape))].reshape(shape)
                vols.append(torch.from_numpy(bits).float().to(self.device))
            self.alphaMask = YinYangAlphaGridMask(self.device, vols[0], vols[1])
        if self.envmap is not None and "envmap.emission" in ckpt:
            self.envmap = EnvironmentMap(h=ckpt["envmap_res_H"], init_strategy="zero", device=self.device)
            self.envmap.load_envmap(emission=ckpt["envmap.emission"], device=self.device)
        self.load_state_dict(ckpt["state_dict"])
        self._scene_cache = None
        if self.coarse_sigma_grid_update_rule == "conv" and self.density_plane_yin[0].is_cuda:
            self.update_coarse_sigma_grid()  # EgoNeRF.py:185-186 (a CPU-resident model only holds weights)
        return ckpt["global_step"]

    def load_state_dict(self, state_dict, strict=True):
        """Values are copied into the existing channel-last parameters (shapes/keys are the reference's)."""
        r = super().load_state_dict(state_dict, strict=strict)
        self._scene_cache = None
        return r


class _WeakOwner:
    """Weak back-reference (keeps MLPRender_Fea out of the module graph cycle / state_dict)."""

    def __init__(self, obj):
        import weakref
        self._r = weakref.ref(obj)

    def __call__(self):
        o = self._r()
        if o is None:
            raise RuntimeError("owning EgoNeRF model was freed")
        return o
