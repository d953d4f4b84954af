"""ctypes binding of include/egonerf_hip.h (the C ABI of libegonerf_hip.so).

This is the only place the product touches native code.  There is no fallback: if the library is
missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import re
from typing import List

# torch first: it bundles its own libamdhip64.so.7; loading ours before torch would bind the process to a
# second HIP runtime copy (/opt/rocm) and torch would then see no device
import torch  # noqa: F401

from .build import LIB

c_float_p = C.POINTER(C.c_float)


class VmField(C.Structure):
    _fields_ = [("plane", (C.c_void_p * 3) * 2), ("line", (C.c_void_p * 3) * 2), ("n_comp", C.c_int32),
                ("res", C.c_int32 * 3)]


class Scene(C.Structure):
    _fields_ = [
        ("center", C.c_float * 3), ("ang_near", C.c_float * 2), ("ang_inv", C.c_float * 2),
        ("r_lut", C.c_void_p), ("n_r_lut", C.c_int32), ("n_r", C.c_int32),
        ("act_softplus", C.c_int32), ("density_shift", C.c_float), ("distance_scale", C.c_float),
        ("density", VmField), ("density_coarse", VmField), ("app", VmField),
        ("basis", C.c_void_p * 2), ("app_dim", C.c_int32),
        ("mlp_w", C.c_void_p * 3), ("mlp_b", C.c_void_p * 3),
        ("mlp_in", C.c_int32), ("mlp_hidden", C.c_int32), ("view_pe", C.c_int32), ("fea_pe", C.c_int32),
        ("packed", C.c_void_p), ("envmap", C.c_void_p), ("envmap_h", C.c_int32), ("mlp_precision", C.c_int32),
        ("occ", C.c_void_p), ("occ_res", C.c_int32 * 3), ("term_eps", C.c_float),
        ("app16", VmField), ("app_f16", C.c_int32), ("n_r_lut_fine", C.c_int32), ("r_lut_fine", C.c_void_p), ("n_r_fine", C.c_int32),
        ("weight_thres", C.c_float), ("occ_cell", C.c_void_p), ("head", C.c_int32),
    ]


def new_scene() -> "Scene":
    """A zero-initialised ego_scene with the opt-in fields that are not 'off' at zero set to off."""
    sc = Scene()
    sc.weight_thres = -1.0
    return sc


class RenderArgs(C.Structure):
    _fields_ = [("n_coarse", C.c_int32), ("n_fine", C.c_int32), ("resampling", C.c_int32),
                ("use_coarse_sample", C.c_int32), ("r_sched", C.c_void_p), ("jitter", C.c_void_p),
                ("u", C.c_void_p), ("near_", C.c_float), ("reserved", C.c_int32), ("z_coarse", C.c_void_p), ("marched", C.c_void_p)]


class VmGrad(C.Structure):
    _fields_ = [("plane", (C.c_void_p * 3) * 2), ("line", (C.c_void_p * 3) * 2)]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                ("lr", C.c_float), ("reserved", C.c_int32)]


class ShadeDump(C.Structure):
    _fields_ = [("x", C.c_void_p), ("h1", C.c_void_p), ("h2", C.c_void_p), ("v", C.c_void_p), ("relu_bits", C.c_void_p), ("fe", C.c_void_p)]


P, I32, I64, F32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SP = C.POINTER(Scene)

# The EGO_ABI_VERSION (include/egonerf_hip.h) the PROTOTYPES below were written against.  load() refuses a library that reports
# another one: a stale libegonerf_hip.so can keep every struct size and still disagree on an argument list (ABI 5 -> 7 inserted
# `normalize` before ego_erp_rays' output pointer), which ctypes would pass through as a wild pointer.
EXPECTED_ABI_VERSION = 17

# name -> (restype, argtypes); mirrors include/egonerf_hip.h one to one
PROTOTYPES = {
    "ego_abi_version": (C.c_int, []),
    "ego_last_error": (C.c_char_p, []),
    "ego_sizeof": (I64, [I32]),
    "ego_selftest_workspace_bytes": (I64, []),
    "ego_selftest": (C.c_int, [P, I64, I32, C.POINTER(C.c_int32), P]),
    "ego_packed_floats": (I64, []),
    "ego_packed_floats_scene": (I64, [SP]),
    "ego_sample_ray_exp": (C.c_int, [P, P, P, F32, I64, I32, P, P, P]),
    "ego_erp_rays": (C.c_int, [I32, I32, I32, I32, C.POINTER(C.c_float), I32, P, P]),
    "ego_copy_out": (C.c_int, [I32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), I32, P]),
    "ego_from_cartesian": (C.c_int, [SP, P, I64, P, P]),
    "ego_normalize_coord": (C.c_int, [SP, P, I64, P, P]),
    "ego_density_feature": (C.c_int, [SP, P, I64, I32, P, P]),
    "ego_app_feature": (C.c_int, [SP, P, I64, P, P]),
    "ego_feature2density": (C.c_int, [SP, P, I64, P, P]),
    "ego_raw2alpha": (C.c_int, [P, P, I64, I32, P, P, P, P]),
    "ego_mlp_fea": (C.c_int, [SP, P, P, I64, P, P]),
    "ego_sample_pdf_merge": (C.c_int, [P, P, P, I64, I32, I32, I32, P, P, P]),
    "ego_envmap_radiance": (C.c_int, [SP, P, I64, P, P]),
    "ego_avgpool_table": (C.c_int, [P, I32, I32, I32, P, P]),
    "ego_avgpool_field": (C.c_int, [C.POINTER(VmField), C.POINTER(VmField), P]),
    "ego_pack_mlp": (C.c_int, [SP, P, P]),
    "ego_pack_mlp_for": (C.c_int, [SP, P, C.c_int32, P]),
    "ego_packed_floats_compat": (C.c_int64, [SP]),
    "ego_pack_mlp_compat": (C.c_int, [SP, P, P]),
    "ego_march_density": (C.c_int, [SP, P, I64, I32, P, P, P, F32, I32, P, P, I32, P, P, P, P, P, P]),
    "ego_shade_kernel_info": (C.c_int, [I32, C.POINTER(C.c_int32), I32]),
    "ego_shade": (C.c_int, [SP, P, P, P, I64, I32, P, P, P, P]),
    "ego_alpha_mask_sample": (C.c_int, [SP, P, I64, P, P]),
    "ego_composite": (C.c_int, [SP, P, P, P, P, P, I64, I32, P, P, P, P, P, P]),
    "ego_shade_composite": (C.c_int, [SP, P, P, P, P, P, I64, I32, P, P, P, P, P, P]),
    "ego_render_forward_folds": (C.c_int32, [SP, I64, I32]),
    "ego_train_packed_floats": (I64, []),
    "ego_pack_train": (C.c_int, [SP, P, P]),
    "ego_train_layout": (C.c_int, [I32, C.POINTER(C.c_int32), I32]),
    "ego_march_backward": (C.c_int, [SP, P, P, I32, P, P, P, P, P, P, P, P, I64, I32, P, P, P]),
    "ego_scatter_density": (C.c_int, [SP, C.POINTER(VmGrad), P, P, I64, I32, P]),
    "ego_scatter_app": (C.c_int, [SP, C.POINTER(VmGrad), P, P, I64, I32, P]),
    "ego_scatter_sorted_workspace_bytes": (C.c_int64, [SP, I64, I32]),
    "ego_scatter_sort": (C.c_int, [SP, P, I64, I32, P, I64, P]),
    "ego_scatter_density_sorted": (C.c_int, [SP, C.POINTER(VmGrad), P, P, I64, I32, P, I64, P]),
    "ego_scatter_app_sorted": (C.c_int, [SP, C.POINTER(VmGrad), P, P, P, P, P, I32, I64, I32, P, I64, P]),
    "ego_envmap_backward": (C.c_int, [SP, P, I32, P, P, P, P, I64, P, P]),
    "ego_shade_backward": (C.c_int, [SP, P, P, P, P, C.POINTER(ShadeDump), P, P, P, P, P, P, I64, I32, P]),
    "ego_sh_render": (C.c_int, [P, P, I64, P, P]),
    "ego_shade_train_generic": (C.c_int, [SP, P, P, I64, I32, P, P, I32, P, P, I32, P, I32, P]),
    "ego_shade_backward_generic": (C.c_int, [SP, P, P, P, P, I32, P, P, I32, P, P, P, P, I32, I64, I32, P]),
    "ego_scatter_generic": (C.c_int, [C.POINTER(VmField), C.POINTER(VmGrad), P, P, I32, I64, I32, P]),
    "ego_weight_grad": (C.c_int, [P, I32, I32, I32, P, P, I32, I32, I32, I32, I64, P, I32, P]),
    "ego_weight_grad_partial_floats": (C.c_int64, []),
    "ego_weight_grad_x": (C.c_int, [P, P, P, P, I32, I32, I64, P, I32, P, I64, P]),
    "ego_weight_grad_det": (C.c_int, [P, I32, I32, I32, P, P, I32, I32, I32, I32, I64, P, I32, P, I64, P]),
    "ego_tv_plane": (C.c_int, [P, I32, I32, I32, F32, P, P, P]),
    "ego_l1_table": (C.c_int, [P, I64, F32, P, P, P]),
    "ego_line_ortho": (C.c_int, [P, I32, I32, F32, P, P, P]),
    "ego_ray_entropy": (C.c_int, [P, I64, I32, I32, P, P, P]),
    "ego_resample_table": (C.c_int, [P, I32, I32, I32, P, P, I32, I32, P, P]),
    "ego_adam_step": (C.c_int, [C.POINTER(AdamTensor), I32, F32, F32, F32, I32, P]),
    "ego_adam_step_graph": (C.c_int, [C.POINTER(AdamTensor), I32, F32, F32, F32, C.c_double, P, P]),
    "ego_rgb_ssim": (C.c_int, [P, P, I32, I32, C.c_double, I32, C.c_double, C.c_double, C.c_double, P, P, P]),
    "ego_render_workspace_bytes": (I64, [I64, C.POINTER(RenderArgs)]),
    "ego_render_forward": (C.c_int, [SP, C.POINTER(RenderArgs), P, I64, P, P, P, P, P, P, P]),
}

_lib = None


def header_symbols() -> List[str]:
    """Every function declared in include/egonerf_hip.h (used by the symbol-export test)."""
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "egonerf_hip.h")
    text = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ego_[a-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise RuntimeError(f"{LIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the EgoNeRF hot path has no CPU fallback)")
    lib = C.CDLL(LIB)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    got = lib.ego_abi_version()
    if got != EXPECTED_ABI_VERSION:
        raise RuntimeError(f"ABI mismatch: {LIB} reports EGO_ABI_VERSION {got}, this binding was written for {EXPECTED_ABI_VERSION}; "
                           "rebuild it (`python -m egonerf_amd.build`)")
    from .build import stale_reason
    why = stale_reason()
    if why and not os.environ.get("EGO_ALLOW_STALE_LIB"):
        raise RuntimeError(f"{LIB} cannot be matched to the sources next to it: {why}; rebuild it (`python -m egonerf_amd.build`) "
                           "or set EGO_ALLOW_STALE_LIB=1 for an experiment build")
    for which, struct in ((0, Scene), (1, RenderArgs), (2, VmField), (3, AdamTensor), (4, ShadeDump)):
        if lib.ego_sizeof(which) != C.sizeof(struct):
            raise RuntimeError(f"ABI mismatch: struct {struct.__name__} is {C.sizeof(struct)} B here, "
                               f"{lib.ego_sizeof(which)} B in {LIB}")
    _lib = lib
    return lib


# ---- run-time self-test of the gather kernels, once per process and device -------------------------------
_SELFTESTED: set = set()
SELFTEST_REPS = int(os.environ.get("EGO_SELFTEST_REPS", "96"))


def selftest(device_index: int, reps: int | None = None, lib: "C.CDLL | None" = None) -> int:
    """Runs ego_selftest on `device_index` (the shipped ego_app_feature / ego_shade kernels, `reps` repetitions on a fixed
    synthetic tile set, bit-compared) and returns the number of mismatching calls; raises on any other error.  `lib`: another
    build of the library (tests hand in the known-faulty form)."""
    lib = lib or load()
    with torch.cuda.device(device_index):
        n = int(lib.ego_selftest_workspace_bytes())
        ws = torch.empty(n // 4 + 64, dtype=torch.float32, device=torch.device("cuda", device_index))  # torch blocks are 512-byte aligned
        bad = C.c_int32(0)
        code = lib.ego_selftest(ws.data_ptr(), n, int(reps or SELFTEST_REPS), C.byref(bad), stream_handle())
        if code != 0 and bad.value == 0:
            raise RuntimeError(f"ego_selftest failed (code {code}): {lib.ego_last_error().decode(errors='replace')}")
    return int(bad.value)


def ensure_selftest(device_index: int) -> None:
    """First use of a device in this process: refuse a library whose gather kernels are not bit-reproducible on it.
    EGO_SKIP_SELFTEST=1 opts out.  ~10 ms; skipped (and retried later) while the current stream is being captured."""
    if device_index in _SELFTESTED:
        return
    if os.environ.get("EGO_SKIP_SELFTEST") == "1":
        _SELFTESTED.add(device_index)
        return
    if torch.cuda.is_current_stream_capturing():
        return
    bad = selftest(device_index)
    if bad:
        raise RuntimeError(f"{LIB}: self-test FAILED on cuda:{device_index}: {bad} calls of the gather kernels returned different bits on identical "
                           f"inputs ({load().ego_last_error().decode(errors='replace')}).  Set EGO_SKIP_SELFTEST=1 only to investigate.")
    _SELFTESTED.add(device_index)


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().ego_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {code}): {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_handle() -> int:
    """Raw hipStream_t of torch's current stream on the CURRENT device; entry points run under `device_guard`, which makes
    the device of their tensors current first."""
    return torch.cuda.current_stream().cuda_stream


def _device_of(args) -> "torch.device | None":
    for a in args:
        if torch.is_tensor(a):
            if a.is_cuda:
                return a.device
        elif isinstance(a, torch.nn.Module):
            p = next(a.parameters(), None)
            if p is not None and p.is_cuda:
                return p.device
        elif isinstance(a, torch.optim.Optimizer):
            for g in a.param_groups:
                for p in g["params"]:
                    if p.is_cuda:
                        return p.device
    return None


def device_guard(fn):
    """Decorator for every entry point that launches kernels (it also runs the library's self-test on the first use of a
    device, `ensure_selftest`): the C ABI takes raw pointers and a raw stream, so the launch
    must happen with the device that owns the pointers current (and on THAT device's current stream).  The owning device is
    the first HIP tensor among the positional arguments (or the first parameter of a module / optimiser argument); when it
    differs from torch's current device the call runs inside `torch.cuda.device(owner)`."""

    @functools.wraps(fn)
    def guarded(*args, **kw):
        dev = _device_of(args) or _device_of(kw.values())
        if dev is not None and dev.index not in _SELFTESTED:
            ensure_selftest(dev.index)
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)

    return guarded
