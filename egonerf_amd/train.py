"""Training step of the render path: forward with kept activations + backward (the autograd of
EgoNeRF.forward that train.py:312-314 runs in the reference).

The differentiable call is a `torch.autograd.Function` whose forward is the same kernel sequence as inference
(coarse march on the pooled tables -> inverse-CDF resampling -> fine march -> shade -> composite; the fine
sample positions are detached like EgoNeRF.py:534) and whose backward runs `ego_march_backward`
(compositing + density, scatter-add into the density tables) and `ego_shade_backward` (MLP / basis data
gradients on the fp16-split MFMA path, scatter-add into the appearance tables).  The basis / MLP weight and
bias gradients come from `ego_weight_grad` (csrc/ego_wgrad.hip: one bf16 hi/lo MFMA pass per layer over the per-sample
buffers the kernels leave behind), un-permuted with the column maps of `ego_train_layout`.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np
import torch

from . import _lib

_LAYOUT_CACHE = {}


def _layout(which: int, n: int, device) -> torch.Tensor:
    key = (which, str(device))
    if key not in _LAYOUT_CACHE:
        buf = (C.c_int32 * n)()
        _lib.check(_lib.load().ego_train_layout(which, buf, n), "ego_train_layout")
        _LAYOUT_CACHE[key] = torch.from_numpy(np.frombuffer(buf, dtype=np.int32).astype(np.int64).copy()).to(device)
    return _LAYOUT_CACHE[key]


_INDEX_CACHE = {}

# The two table scatters (VALU / atomic bound) run on a side stream next to the shade backward and the weight-gradient passes
# (HBM bound).  Every buffer the side stream reads stays referenced until the main stream has waited for it: a tensor freed
# earlier goes back to the main stream's allocator pool and the next main-stream allocation may overwrite it while the side
# stream still reads (the intermittent wrong appearance gradient of the first attempt).  EGO_TRAIN_SIDE_STREAM=0
# serialises everything on one stream.
SIDE_STREAM_SCATTER = __import__("os").environ.get("EGO_TRAIN_SIDE_STREAM", "1") != "0"
# r06: d(basis) rides along in the sorted appearance scatter's walk (plane value x line value = v is in its registers), so the forward
# dumps no v and the d(basis) weight-gradient pass is gone; EGO_TRAIN_WALK_BASIS=0 keeps the dump + ego_weight_grad form
WALK_BASIS = __import__("os").environ.get("EGO_TRAIN_WALK_BASIS", "1") != "0"
# r06: ... and re-derives dv = basis^T dfe there too (27 slot gradients in, 48 channels out per plane), so ego_shade_backward writes no dv
# (576 B per sample) for the scatter to read back; EGO_TRAIN_WALK_DV=0 keeps the dv hand-over
WALK_DV = __import__("os").environ.get("EGO_TRAIN_WALK_DV", "1") != "0"
DUMP_X = __import__("os").environ.get("EGO_TRAIN_DUMP_X", "0") != "0"   # keep the forward's x dump (rounds 1-4) instead of re-deriving x for d(W1)
_SIDE_STREAMS = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


# Optional per-call timing (bench.py): when KERNEL_MARKS is a list, an event is recorded on the current stream after every library
# call of the step and appended as (name, event); the interval between two consecutive marks is that call's kernel time plus the
# torch glue (zero fills, index scatters) queued in front of it.  Meaningful with SIDE_STREAM_SCATTER = False (one stream).
KERNEL_MARKS = None


_DEBUG_SYNC = __import__("os").environ.get("EGO_TRAIN_DEBUG_SYNC", "0") != "0"   # debugging: synchronise and print after every library call


def _chk(code: int, what: str) -> None:
    _lib.check(code, what)
    if _DEBUG_SYNC and not torch.cuda.is_current_stream_capturing():
        print("[ego train] queued", what, flush=True)
        torch.cuda.synchronize()
        print("[ego train] done  ", what, flush=True)
    if KERNEL_MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        KERNEL_MARKS.append((what, ev))


def mark(what: str) -> None:
    """An extra mark (e.g. "begin") for callers that collect KERNEL_MARKS."""
    if KERNEL_MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        KERNEL_MARKS.append((what, ev))


# Weight-gradient products of one step land in ONE zero-filled buffer [352][160] (rows: do^T h2 | dh2^T h1 | dh1^T x | dfe^T v) and
# ONE gather turns it into the eight reference-shaped gradients (basis yin / yang, w1, b1, w2, b2, w3, b3): two small kernels
# instead of a zero fill per product and a fill + an index_put per gradient (22 launches of ~5 us each on the critical path).
_G_ROWS = dict(G3=(0, 32), G2=(32, 160), G1=(160, 288), Gb=(288, 352))
_G_LD = 160


def _grad_plan(model, device, basis_reference_columns: bool = False):
    """(flat gather index, split sizes, shapes, pad column) derived once from the layouts (no host synchronisation per step).
    basis_reference_columns: the Gb block's columns are already plane x 48 + channel (ego_scatter_app_sorted's gbasis) instead of the
    v dump's column order (ego_weight_grad over ego_shade_dump.v)."""
    key = (str(device), bool(basis_reference_columns))
    if key not in _INDEX_CACHE:
        hid, xmap, fmap, vmap = (_layout(w, n, "cpu") for w, n in ((1, 128), (0, 160), (2, 32), (3, 144)))
        x_sel = (xmap >= 0).nonzero().flatten()
        f_sel = (fmap >= 0).nonzero().flatten()
        x_cols, f_rows, pad = xmap[x_sel], fmap[f_sel], int((xmap < 0).nonzero()[0])
        src = torch.arange(352 * _G_LD).view(352, _G_LD)   # every element of the product buffer, by flat position
        G3, G2, G1, Gb = (src[a:b] for a, b in (_G_ROWS[k] for k in ("G3", "G2", "G1", "Gb")))
        mlp = model.renderModule.mlp
        neg = lambda *shape: torch.full(shape, -1, dtype=torch.int64)
        gw3 = neg(*mlp[4].weight.shape); gw3[:, hid] = G3[:3, :128]
        gb3 = G3[:3, 128].clone()
        gw2 = neg(*mlp[2].weight.shape); gw2[hid[:, None], hid[None, :]] = G2[:, :128]
        gb2 = neg(*mlp[2].bias.shape); gb2[hid] = G2[:, 128]
        gw1 = neg(*mlp[0].weight.shape); gw1[hid[:, None], x_cols[None, :]] = G1[:, x_sel]
        gb1 = neg(*mlp[0].bias.shape); gb1[hid] = G1[:, pad]
        gbasis = []
        for g in range(2):
            gb = neg(model.app_dim, 144)
            cols = torch.arange(144) if basis_reference_columns else vmap
            gb[f_rows[:, None], cols[None, :]] = Gb[32 * g: 32 * g + 32, :144][f_sel]
            gbasis.append(gb)
        parts = gbasis + [gw1, gb1, gw2, gb2, gw3, gb3]
        flat = torch.cat([p.reshape(-1) for p in parts])
        assert int(flat.min()) >= 0, "every weight / bias gradient element has a source in the product buffer"
        _INDEX_CACHE[key] = (flat.to(device), [p.numel() for p in parts], [tuple(p.shape) for p in parts], pad)
    return _INDEX_CACHE[key]


def _zeros_like_many(tensors, zero: bool = True):
    """One zero-filled flat buffer carved into tensors with the shapes AND strides of `tensors` (dense, e.g. channel-last
    parameters): one fill kernel instead of one per gradient.  zero=False: uninitialised (the sorted scatters write every texel)."""
    total = sum(t.numel() for t in tensors)
    flat = (torch.zeros if zero else torch.empty)(total, device=tensors[0].device, dtype=tensors[0].dtype)
    out, off = [], 0
    for t in tensors:
        out.append(torch.as_strided(flat, t.shape, t.stride(), off))
        off += t.numel()
    return out


def _grad_struct(tensors: List[torch.Tensor]) -> "_lib.VmGrad":
    """tensors: [plane_yin x3, line_yin x3, plane_yang x3, line_yang x3] gradient tables (channel-last memory)."""
    g = _lib.VmGrad()
    for gi in range(2):
        for i in range(3):
            g.plane[gi][i] = tensors[gi * 6 + i].data_ptr()
            g.line[gi][i] = tensors[gi * 6 + 3 + i].data_ptr()
    return g


def table_params(model, kind: str) -> List[torch.nn.Parameter]:
    out = []
    for g in ("yin", "yang"):
        out += list(getattr(model, f"{kind}_plane_{g}")) + list(getattr(model, f"{kind}_line_{g}"))
    return out


def head_is_tuned(model) -> bool:
    """True when the appearance head has the shape the MFMA kernels (forward dumps, ego_shade_backward, ego_scatter_app) are built for:
    48 components, app_dim 27, MLP_Fea 150 -> 128 -> 128 -> 3 with view_pe = fea_pe = 2.  (The density field's component count is
    independent: 16 takes ego_scatter_density, anything else ego_scatter_generic.)"""
    return model.head_is_tuned


def _generic_head_sorted_app(model, sort_ws, M: int) -> bool:
    """A head of another shape over 48-component appearance tables: ego_shade_backward_generic writes the blocked dv and the sorted walk
    scatters it (the step's sort exists whenever the density field has its tuned shape)."""
    return (sort_ws is not None and model.app_n_comp[0] == 48 and __import__("os").environ.get("EGO_SORTED_WALK", "1") != "0"
            and (M + 31) // 32 * 32 * 144 < 2 ** 30)


def differentiable_params(model) -> List[torch.nn.Parameter]:
    """Fixed order: 12 density tables, 12 appearance tables, basis yin/yang, mlp (w0,b0,w1,b1,w2,b2)."""
    head = [model.basis_mat_yin.weight, model.basis_mat_yang.weight]
    if model.shadingMode != "RGB":   # RGBRender has no parameters (tensorBase.py:37-39)
        m = model.renderModule.mlp
        head += [m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias]
    return table_params(model, "density") + table_params(model, "app") + head


class RenderFunction(torch.autograd.Function):
    @staticmethod
    @_lib.device_guard
    def forward(ctx, model, rays, opts, *params):
        """params = differentiable_params(model) (+ envmap.emission last when the model has an envmap)."""
        lib, st = _lib.load(), _lib.stream_handle()
        ctx.set_materialize_grads(False)
        dev = rays.device
        # training gathers from the fp32 parameters (the backward re-gathers from them), shades every sample like EgoNeRF.forward (the
        # appearance skip is an inference option) and keeps all three fp16 terms of every product, whatever model.mlp_precision says
        if model._mlp_precision == "f32" and not model.train_fp32_head and not getattr(model, "_warned_f32_training", False):
            # mlp_precision is an INFERENCE switch; before r05 a differentiable call with "f32" raised, now it would silently
            # train through split-fp16 products and half-precision activation dumps - say so once, and say what the real switch is
            import warnings
            warnings.warn("EgoNeRF.mlp_precision = 'f32' applies to inference only: the differentiable path runs the f16x3 training kernels "
                          "(fp16-split MFMA products, half-precision activation dumps).  For fp32 activations and operands in training set "
                          "model.train_fp32_head = True (or EGO_TRAIN_FP32=1): the parity mode, several times slower.", RuntimeWarning, stacklevel=3)
            model._warned_f32_training = True
        sc = model.scene(training=True)
        N = rays.shape[0]
        n_coarse, n_fine = opts["n_coarse"], opts["n_fine"]
        resampling, use_coarse = opts["resampling"], opts["use_coarse_sample"]
        jitter, u = opts["jitter"], opts["u"]
        zc_in = opts.get("z_coarse")  # explicit first-pass distances (exp_sampling=False) instead of schedule + jitter
        near = float(model.near_far[0])
        sched = model._sched(n_coarse, dev)
        S = (n_coarse + n_fine if use_coarse else n_fine) if resampling else n_coarse
        f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        z, alpha, weight, sigma, bg = f(N, S), f(N, S + int(model.envmap is not None)), f(N, S), f(N, S), f(N)
        coords = f(N, S, 4)
        astride = alpha.shape[1]
        if resampling:
            zc, wc = (f(N, n_coarse) if zc_in is None else zc_in), f(N, n_coarse)
            _chk(lib.ego_march_density(sc, rays.data_ptr(), N, n_coarse, _lib.ptr(zc_in), None if zc_in is not None else sched.data_ptr(),
                                             None if zc_in is not None else _lib.ptr(jitter), near, 1,
                                             None if zc_in is not None else zc.data_ptr(), None, 0, wc.data_ptr(), None, None, None, None, st),
                       "ego_march_density")
            _chk(lib.ego_sample_pdf_merge(zc.data_ptr(), wc.data_ptr(), _lib.ptr(u), N, n_coarse, n_fine, int(use_coarse),
                                                z.data_ptr(), None, st), "ego_sample_pdf_merge")
            # fine pass: full tables and the full-resolution r grid (bit 1; EgoNeRF.py:546 normalises without `downsample`)
            _chk(lib.ego_march_density(sc, rays.data_ptr(), N, S, z.data_ptr(), None, None, near, 2, None, alpha.data_ptr(),
                                             astride, weight.data_ptr(), bg.data_ptr(), coords.data_ptr(), sigma.data_ptr(), None, st),
                       "ego_march_density")
        else:
            _chk(lib.ego_march_density(sc, rays.data_ptr(), N, S, _lib.ptr(zc_in), None if zc_in is not None else sched.data_ptr(),
                                             None if zc_in is not None else _lib.ptr(jitter), near, 0,
                                             z.data_ptr(), alpha.data_ptr(), astride, weight.data_ptr(), bg.data_ptr(),
                                             coords.data_ptr(), sigma.data_ptr(), None, st), "ego_march_density")
        M = N * S
        rgb = f(N, S, 3)
        # Deterministic table gradients (model.deterministic_scatter, the default): the step's samples are binned by texel cell once, from
        # the coordinates the march just wrote - on the side stream, next to the dumping shade forward - and the backward's two scatters
        # then write every gradient texel once, in a fixed order (csrc/ego_scatter_sorted.hip).  Tuned table shapes only (16 / 48
        # components); anything else keeps the atomic scatters.
        sort_ws = None
        if model.deterministic_scatter and (model.density_n_comp[0] == 16 or (head_is_tuned(model) and not model.train_fp32_head)):
            nbytes = lib.ego_scatter_sorted_workspace_bytes(sc, N, S)
            if nbytes <= 0:
                raise RuntimeError("ego_scatter_sorted_workspace_bytes rejected the scene: " + lib.ego_last_error().decode())
            sort_ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev) if SIDE_STREAM_SCATTER else None
            if side is None:
                _chk(lib.ego_scatter_sort(sc, coords.data_ptr(), N, S, sort_ws.data_ptr(), nbytes, st), "ego_scatter_sort")
            else:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    _chk(lib.ego_scatter_sort(sc, coords.data_ptr(), N, S, sort_ws.data_ptr(), nbytes, _lib.stream_handle()), "ego_scatter_sort")
                # a forward whose backward never runs must not hand these blocks back to the main stream's pool while the sort is in flight
                sort_ws.record_stream(side)
                coords.record_stream(side)
        head_tuned = head_is_tuned(model) and not model.train_fp32_head
        compat = None
        if head_is_tuned(model) and not head_tuned:
            # parity mode (model.train_fp32_head / EGO_TRAIN_FP32=1): the tuned head trained through the fp32 compatibility kernels -
            # fp32 activations, fp32 dumps, fp32 weight-gradient operands like the reference's autograd - on a copy of the scene whose
            # packed blob is the compatibility layout (the tuned path keeps x / h1 / h2 / dh1 / dh2 as halves: DESIGN.md 4.2)
            sc = _lib.Scene.from_buffer_copy(sc)
            compat = f(lib.ego_packed_floats_compat(sc))
            _chk(lib.ego_pack_mlp_compat(sc, compat.data_ptr(), st), "ego_pack_mlp_compat")
            sc.packed = compat.data_ptr()
        if head_tuned:
            Mp = (M + 31) // 32 * 32  # the dumps are tile-blocked (include/egonerf_hip.h, ego_shade_dump): whole tiles
            f16 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float16)   # x / h1 / h2: halves in the kernels' operand order
            # x (the 160-column MLP input) is not dumped: d(W1) re-derives it from the feature slots `fe` and the view direction
            # (ego_weight_grad_x: bit-identical operands, 0.67 GB less written and 0.4 GB less read per 8192 x 256 step); DUMP_X keeps the dump
            walk_basis = bool(WALK_BASIS and sort_ws is not None and __import__("os").environ.get("EGO_SORTED_WALK", "1") != "0" and Mp * 144 < 2 ** 30)
            dump = dict(x=f16(Mp, 160) if DUMP_X else None, h1=f16(Mp, 128), h2=f16(Mp, 128), v=None if walk_basis else f(Mp, 144),
                        relu_bits=torch.empty(Mp // 32, 2, 64, 2, device=dev, dtype=torch.int32), fe=f(Mp // 32, 4, 64, 4))
            ds = _lib.ShadeDump(*(_lib.ptr(dump[k]) for k in ("x", "h1", "h2", "v", "relu_bits", "fe")))
            _chk(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), coords.data_ptr(), N, S, rgb.data_ptr(), C.byref(ds), None, st), "ego_shade")
        else:
            # any other model shape (opt.py:87-100): fp32 compatibility kernels over row-major dumps, padded to whole 160-column
            # blocks (what ego_weight_grad multiplies at a time) with at least one zero column left for the bias gradients
            in_c = model.head_in_mlpC
            ldx = (in_c // _G_LD + 1) * _G_LD
            z0 = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.float32)
            if model.shadingMode == "RGB":   # no network: only the plane x line products are kept (they feed the basis gradient)
                dump = dict(x=None, h1=None, h2=None, v=z0(M, _G_LD))
            else:
                dump = dict(x=z0(M, ldx), h1=z0(M, _G_LD), h2=z0(M, _G_LD), v=z0(M, _G_LD))
            _chk(lib.ego_shade_train_generic(sc, rays.data_ptr(), coords.data_ptr(), N, S, rgb.data_ptr(), _lib.ptr(dump["x"]), ldx,
                                             _lib.ptr(dump["h1"]), _lib.ptr(dump["h2"]), _G_LD, dump["v"].data_ptr(), _G_LD, st),
                 "ego_shade_train_generic")
        rgb_map, depth, raw = f(N, 3), f(N), f(N, 3)
        has_env = model.envmap is not None
        bg_map = f(N, 3) if has_env else None
        env_map = f(N, 3) if has_env else None
        _chk(lib.ego_composite(sc, rays.data_ptr(), z.data_ptr(), weight.data_ptr(), bg.data_ptr(), rgb.data_ptr(), N, S,
                                     rgb_map.data_ptr(), depth.data_ptr(), _lib.ptr(bg_map), _lib.ptr(env_map), raw.data_ptr(), st),
                   "ego_composite")
        ctx.model, ctx.N, ctx.S, ctx.head_tuned, ctx.compat, ctx.sort_ws = model, N, S, head_tuned, compat, sort_ws
        ctx.saved = dict(z=z, alpha=alpha, weight=weight, sigma=sigma, bg=bg, coords=coords, rgb=rgb, raw=raw, env=env_map, rays=rays,
                         **dump)
        # depth is computed under no_grad in the reference (EgoNeRF.py:595-598); one call: a second would replace the first
        if has_env:
            ctx.mark_non_differentiable(depth, bg_map, env_map)
            return rgb_map, depth, alpha, bg_map, env_map
        ctx.mark_non_differentiable(depth)
        return rgb_map, depth, alpha

    @staticmethod
    @_lib.device_guard
    def backward(ctx, g_rgb, _g_depth=None, g_alpha=None, *_unused):
        lib, st = _lib.load(), _lib.stream_handle()
        model, N, S, sv = ctx.model, ctx.N, ctx.S, ctx.saved
        dev = sv["z"].device
        M = N * S
        sc = model.scene(training=True)
        if ctx.compat is not None:   # the forward's compatibility blob (fp32 parity mode of the tuned head)
            sc = _lib.Scene.from_buffer_copy(sc)
            sc.packed = ctx.compat.data_ptr()
        g_rgb = torch.zeros(N, 3, device=dev) if g_rgb is None else g_rgb.contiguous().float()
        astride = sv["alpha"].shape[1]
        if g_alpha is not None:  # ray_entropy_loss (train.py:306-309); the envmap's ones column takes no gradient
            g_alpha = g_alpha.contiguous().float()
            assert g_alpha.shape == sv["alpha"].shape
        dens, app = table_params(model, "density"), table_params(model, "app")
        # keeps the channel-last strides of the parameters; the sorted scatters store every texel, the atomic ones add into zeros
        sorted_dens = ctx.sort_ws is not None and model.density_n_comp[0] == 16
        sorted_app = ctx.sort_ws is not None and (ctx.head_tuned or _generic_head_sorted_app(model, ctx.sort_ws, N * S))
        g_dens, g_app = _zeros_like_many(dens, zero=not sorted_dens), _zeros_like_many(app, zero=not sorted_app)
        for p, g in zip(dens + app, g_dens + g_app):
            assert g.stride() == p.stride()
        f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        dc, dfeat = f(N, S, 3), f(N, S)
        _chk(lib.ego_march_backward(sc, sv["z"].data_ptr(), sv["alpha"].data_ptr(), astride, sv["weight"].data_ptr(),
                                          sv["sigma"].data_ptr(), sv["bg"].data_ptr(), sv["rgb"].data_ptr(), g_rgb.data_ptr(),
                                          _lib.ptr(g_alpha), sv["raw"].data_ptr(), _lib.ptr(sv["env"]), N, S, dc.data_ptr(),
                                          dfeat.data_ptr(), st), "ego_march_backward")
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if SIDE_STREAM_SCATTER else None
        gd = _grad_struct(g_dens)

        def on_side(fn):
            """Run fn(stream handle) on the side stream once everything queued on the main stream so far has finished."""
            if side is None:
                return fn(st)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fn(_lib.stream_handle())

        try:
            return RenderFunction._backward_body(ctx, lib, st, model, N, S, sv, dev, M, sc, g_rgb, astride, g_dens, g_app, dc, dfeat, gd,
                                                 main, side, on_side, f)
        finally:
            # Whatever happens in between (a failing _lib.check raises), the main stream must have waited for the side stream before
            # dfeat / dv / coords / the gradient buffer go back to the main stream's allocator pool: the side-stream kernels may still
            # be reading or writing them.
            if side is not None:
                main.wait_stream(side)
            ctx.saved = None

    @staticmethod
    def _backward_body(ctx, lib, st, model, N, S, sv, dev, M, sc, g_rgb, astride, g_dens, g_app, dc, dfeat, gd, main, side, on_side, f):
        ws = ctx.sort_ws
        if model.density_n_comp[0] == 16 and ws is not None:
            on_side(lambda s_: _chk(lib.ego_scatter_density_sorted(sc, C.byref(gd), sv["coords"].data_ptr(), dfeat.data_ptr(), N, S, ws.data_ptr(),
                                                                   ws.numel(), s_), "ego_scatter_density_sorted"))
        elif model.density_n_comp[0] == 16:
            on_side(lambda s_: _chk(lib.ego_scatter_density(sc, C.byref(gd), sv["coords"].data_ptr(), dfeat.data_ptr(), N, S, s_),
                                          "ego_scatter_density"))
        else:
            on_side(lambda s_: _chk(lib.ego_scatter_generic(C.byref(sc.density), C.byref(gd), sv["coords"].data_ptr(), dfeat.data_ptr(), 0, N, S, s_),
                                          "ego_scatter_generic(density)"))
        if not ctx.head_tuned:
            return RenderFunction._backward_generic_head(ctx, lib, st, model, N, S, sv, dev, M, sc, g_rgb, g_dens, g_app, dc, main, side, on_side, f)
        tp = f(lib.ego_train_packed_floats())
        _chk(lib.ego_pack_train(sc, tp.data_ptr(), st), "ego_pack_train")
        Mp = (M + 31) // 32 * 32
        # dh2 / dh1: scaled fp16 in the kernel's own operand order + one power of two per sample (include/egonerf_hip.h); dv: blocked fp32
        half = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float16)
        walk_basis = ws is not None and sv["v"] is None
        walk_dv = walk_basis and WALK_DV
        dh2, dh1, dh_scale, dfe, dv = half(Mp, 128), half(Mp, 128), f(2, Mp), f(M, 32), (None if walk_dv else f(Mp, 144))
        dv_absmax = f(1)   # max |dv| (walk_dv: max |dfe|), found by the backward while it holds the values: the sorted scatter's fixed-point unit comes from it
        ds = _lib.ShadeDump(*(_lib.ptr(sv[k]) for k in ("x", "h1", "h2", "v", "relu_bits", "fe")))
        _chk(lib.ego_shade_backward(sc, tp.data_ptr(), sv["coords"].data_ptr(), dc.data_ptr(), sv["rgb"].data_ptr(), C.byref(ds),
                                          dh2.data_ptr(), dh1.data_ptr(), dh_scale.data_ptr(), dfe.data_ptr(), _lib.ptr(dv), dv_absmax.data_ptr(), N, S, st),
             "ego_shade_backward")
        ga = _grad_struct(g_app)
        # the product buffer exists (zero-filled, on the main stream) BEFORE the side stream is allowed to write its d(basis) block into it
        Gall = torch.zeros(352, _G_LD, device=dev)
        if ws is not None:   # (this is the tuned-head path: sorted_app above)
            Gb_ptr = Gall[_G_ROWS["Gb"][0]:].data_ptr() if walk_basis else None
            on_side(lambda s_: _chk(lib.ego_scatter_app_sorted(sc, C.byref(ga), sv["coords"].data_ptr(), _lib.ptr(dv), dv_absmax.data_ptr(),
                                                               dfe.data_ptr() if walk_basis else None, Gb_ptr, _G_LD if walk_basis else 0, N, S,
                                                               ws.data_ptr(), ws.numel(), s_), "ego_scatter_app_sorted"))
        else:
            on_side(lambda s_: _chk(lib.ego_scatter_app(sc, C.byref(ga), sv["coords"].data_ptr(), dv.data_ptr(), N, S, s_), "ego_scatter_app"))
        if side is None:
            del dv
        # ---- weight gradients: one pass of ego_weight_grad (bf16 hi/lo MFMA over transposed LDS tiles, bias gradients from a
        # ones column) per layer over the dumped buffers into one product buffer, then one gather un-permutes the lane-order
        # columns into the reference-shaped gradients ----
        do = dc.view(M, 3)  # now d(pre-sigmoid)
        gidx, gsizes, gshapes, pad = _grad_plan(model, dev, walk_basis)  # pad: a padding column of the x dump (holds zeros) doubles as the ones column
        # deterministic mode (model.deterministic_scatter): per-workgroup partial products added in a fixed order instead of float atomics
        det = model.deterministic_scatter
        part = f(lib.ego_weight_grad_partial_floats()) if det else None

        def wgrad(which, A, ca, a_layout, B, cb, ones_col, a_scale=None):
            G = Gall[_G_ROWS[which][0]:_G_ROWS[which][1]]
            b_layout = 2 if B.dtype == torch.float16 else 1
            _chk(lib.ego_weight_grad_det(A.data_ptr(), A.shape[1], ca, a_layout, _lib.ptr(a_scale), B.data_ptr(), B.shape[1], cb, b_layout, ones_col, M,
                                         G.data_ptr(), _G_LD, _lib.ptr(part), 0 if part is None else part.numel(), st), "ego_weight_grad")

        wgrad("G3", do, 3, 0, sv["h2"], 128, 128)
        wgrad("G2", dh2, 128, 2, sv["h1"], 128, 128, dh_scale[0])
        if sv["x"] is not None:
            wgrad("G1", dh1, 128, 2, sv["x"], 160, pad, dh_scale[1])
        else:   # the layer-1 input re-derived from the feature slots and the rays' directions inside the product's B-tile fetch
            G1 = Gall[_G_ROWS["G1"][0]:_G_ROWS["G1"][1]]
            _chk(lib.ego_weight_grad_x(dh1.data_ptr(), dh_scale[1].data_ptr(), sv["fe"].data_ptr(), sv["rays"].data_ptr(), S, pad, M, G1.data_ptr(), _G_LD,
                                       _lib.ptr(part), 0 if part is None else part.numel(), st), "ego_weight_grad")
        if walk_basis:
            main.wait_stream(side) if side is not None else None   # the walk's k_basis_reduce wrote Gall's Gb block on the side stream
        else:
            wgrad("Gb", dfe, 64, 3, sv["v"], 144, -1, sv["coords"].view(M, 4))   # 32 stored columns, routed to the yin / yang block by coords.w
        wg = [t.view(shp) for t, shp in zip(Gall.view(-1).index_select(0, gidx).split(gsizes), gshapes)]
        grads = g_dens + g_app + wg  # wg: basis yin, basis yang, w1, b1, w2, b2, w3, b3 (differentiable_params order)
        if sv["env"] is not None:
            g_em = torch.zeros_like(model.envmap.emission)
            rays = sv["rays"]
            _chk(lib.ego_envmap_backward(sc, rays.data_ptr() + 12, 6, g_rgb.data_ptr(), sv["raw"].data_ptr(), sv["bg"].data_ptr(),
                                               sv["env"].data_ptr(), N, g_em.data_ptr(), st), "ego_envmap_backward")
            grads.append(g_em)
        if side is not None:
            main.wait_stream(side)  # the table gradients are complete; dv / dfeat / coords may be released from here on
        return (None, None, None, *grads)


    @staticmethod
    def _backward_generic_head(ctx, lib, st, model, N, S, sv, dev, M, sc, g_rgb, g_dens, g_app, dc, main, side, on_side, f):
        """Backward of the appearance head for any model shape the compatibility kernels support (row-major buffers; the weight
        gradients are A^T B products in 160-column blocks, already in the reference's [out][in] orientation)."""
        hid, in_c, ncol, D = model.head_hidden, model.head_in_mlpC, 3 * model.app_n_comp[0], model.app_dim
        rgb_head = model.shadingMode == "RGB"
        ldx = _G_LD if rgb_head else sv["x"].shape[1]
        # another head over the SHIPPED table shape: dv in the tuned scatters' blocked layout, and the sorted walk takes the table gradients
        # (deterministic, 0.75 ms where the any-shape atomic scatter takes 9.8)
        ws = ctx.sort_ws
        app_sorted = _generic_head_sorted_app(model, ws, M)
        dh2, dh1, dfe = (None if rgb_head else f(M, hid)), (None if rgb_head else f(M, hid)), f(M, 64)
        dv = f((M + 31) // 32 * 32, 144) if app_sorted else f(M, _G_LD)
        _chk(lib.ego_shade_backward_generic(sc, sv["coords"].data_ptr(), dc.data_ptr(), sv["rgb"].data_ptr(), _lib.ptr(sv["x"]), ldx,
                                            _lib.ptr(sv["h1"]), _lib.ptr(sv["h2"]), _G_LD, _lib.ptr(dh2), _lib.ptr(dh1), dfe.data_ptr(),
                                            dv.data_ptr(), 0 if app_sorted else _G_LD, N, S, st), "ego_shade_backward_generic")
        ga = _grad_struct(g_app)
        if app_sorted:
            on_side(lambda s_: _chk(lib.ego_scatter_app_sorted(sc, C.byref(ga), sv["coords"].data_ptr(), dv.data_ptr(), None, None, None, 0, N, S,
                                                               ws.data_ptr(), ws.numel(), s_), "ego_scatter_app_sorted(generic head)"))
        else:
            on_side(lambda s_: _chk(lib.ego_scatter_generic(C.byref(sc.app), C.byref(ga), sv["coords"].data_ptr(), dv.data_ptr(), _G_LD, N, S, s_),
                                    "ego_scatter_generic(app)"))
        hp = (hid + 31) // 32 * 32
        n_chunks = ldx // _G_LD

        part = f(lib.ego_weight_grad_partial_floats()) if model.deterministic_scatter else None

        def product(A, lda, ca, B_ptr, ldb, ones_col, rows):
            G = torch.zeros(rows, _G_LD, device=dev)
            _chk(lib.ego_weight_grad_det(A.data_ptr(), lda, ca, 0, None, B_ptr, ldb, _G_LD, 0, ones_col, M, G.data_ptr(), _G_LD, _lib.ptr(part),
                                         0 if part is None else part.numel(), st), "ego_weight_grad")
            return G

        Gb = product(dfe, 64, 64, sv["v"].data_ptr(), _G_LD, -1, 64)                      # [yin | yang] feature gradients ^T v
        wg = [Gb[0:D, :ncol].contiguous(), Gb[32:32 + D, :ncol].contiguous()]
        if not rgb_head:
            G3 = product(dc.view(M, 3), 3, 3, sv["h2"].data_ptr(), _G_LD, hid, 32)          # do^T [h2 | 1]
            G2 = product(dh2, hid, hid, sv["h1"].data_ptr(), _G_LD, hid, hp)                  # dh2^T [h1 | 1]
            G1 = []
            for c in range(n_chunks):
                c0 = c * _G_LD
                ones = in_c - c0 if c0 <= in_c < c0 + _G_LD else -1                          # the first zero-padding column doubles as the ones column
                G1.append(product(dh1, hid, hid, sv["x"].data_ptr() + 4 * c0, ldx, ones, hp))
            G1 = torch.cat(G1, dim=1)
            wg += [G1[:hid, :in_c].contiguous(), G1[:hid, in_c].contiguous(), G2[:hid, :hid].contiguous(), G2[:hid, hid].contiguous(),
                   G3[:3, :hid].contiguous(), G3[:3, hid].contiguous()]
        grads = g_dens + g_app + wg
        if sv["env"] is not None:
            g_em = torch.zeros_like(model.envmap.emission)
            rays = sv["rays"]
            _chk(lib.ego_envmap_backward(sc, rays.data_ptr() + 12, 6, g_rgb.data_ptr(), sv["raw"].data_ptr(), sv["bg"].data_ptr(),
                                         sv["env"].data_ptr(), N, g_em.data_ptr(), st), "ego_envmap_backward")
            grads.append(g_em)
        if side is not None:
            main.wait_stream(side)
        return (None, None, None, *grads)


class EnvRadianceFunction(torch.autograd.Function):
    """EnvironmentMap.get_radiance with a gradient to `emission` (the reference's envmap pre-training, train.py:218-236)."""

    @staticmethod
    @_lib.device_guard
    def forward(ctx, emission, dirs):
        lib, st = _lib.load(), _lib.stream_handle()
        sc = _lib.new_scene()
        sc.envmap, sc.envmap_h = emission.data_ptr(), emission.shape[2]
        out = torch.empty(dirs.shape[0], 3, device=dirs.device)
        _lib.check(lib.ego_envmap_radiance(sc, dirs.data_ptr(), dirs.shape[0], out.data_ptr(), st), "ego_envmap_radiance")
        ctx.save_for_backward(dirs, out)
        ctx.shape = emission.shape
        return out

    @staticmethod
    @_lib.device_guard
    def backward(ctx, g):
        lib, st = _lib.load(), _lib.stream_handle()
        dirs, out = ctx.saved_tensors
        N = dirs.shape[0]
        sc = _lib.new_scene()
        sc.envmap_h = ctx.shape[2]
        g_em = torch.zeros(ctx.shape, device=dirs.device)
        ones, inside = torch.ones(N, device=dirs.device), torch.zeros(N, 3, device=dirs.device)
        _lib.check(lib.ego_envmap_backward(sc, dirs.data_ptr(), 3, g.contiguous().float().data_ptr(), inside.data_ptr(), ones.data_ptr(),
                                           out.data_ptr(), N, g_em.data_ptr(), st), "ego_envmap_backward")
        return g_em, None


def render_train(model, rays, n_coarse, n_fine=0, resampling=False, use_coarse_sample=True, jitter: Optional[torch.Tensor] = None,
                 u: Optional[torch.Tensor] = None, z_coarse: Optional[torch.Tensor] = None):
    """Differentiable EgoNeRF.forward (is_train semantics) -> (rgb_map, depth, bg_map|None, env_map|None, alpha)."""
    opts = dict(n_coarse=int(n_coarse), n_fine=int(n_fine), resampling=bool(resampling), use_coarse_sample=bool(use_coarse_sample),
                jitter=jitter, u=u, z_coarse=z_coarse)
    params = differentiable_params(model)
    if model.envmap is not None:
        params = params + [model.envmap.emission]
    out = RenderFunction.apply(model, rays, opts, *params)
    if model.envmap is not None:
        rgb_map, depth, alpha, bg_map, env_map = out
        return rgb_map, depth, bg_map, env_map, alpha
    rgb_map, depth, alpha = out
    return rgb_map, depth, None, None, alpha


class TrainSchedule:
    """Device-side iteration clock for loss terms whose weights change from iteration to iteration.  train.py:295-309 multiplies
    TV_weight_density / TV_weight_app (while `iteration < iter_ignore_TV`) and entropy_weight (once `iteration > iter_ignore_entropy`)
    by lr_factor in every iteration in which the term is active; a Python float captured into a hipGraph would freeze at its value
    of the capture.  `it` is a float64 device scalar holding the index of the CURRENT iteration (GraphedTrainStep advances it inside
    the graph), and `decayed()` evaluates the reference's running product as a device expression."""

    def __init__(self, device, start_iteration: int = 0):
        self.first = int(start_iteration)
        self.it = torch.full((), float(start_iteration), dtype=torch.float64, device=device)

    def decayed(self, w0: float, factor: float, active_before: Optional[int] = None, active_after: Optional[int] = None) -> torch.Tensor:
        """The weight the reference's loop holds in the current iteration (float32 device scalar; 0 when the term is gated off):
        `active_before=K`: term active while iteration < K, `w *= factor` in every active iteration before use (the TV terms,
        train.py:295-304); `active_after=K`: active once iteration > K (the ray-entropy term, train.py:306-309)."""
        it = self.it
        if (active_before is None) == (active_after is None):
            raise ValueError("TrainSchedule.decayed: exactly one of active_before / active_after")
        if active_before is not None:
            n, gate = it - (self.first - 1), it < float(active_before)
        else:
            n, gate = it - float(max(int(active_after), self.first - 1)), it > float(active_after)
        w = float(w0) * torch.pow(torch.full_like(it, float(factor)), n)
        return torch.where(gate, w, torch.zeros_like(w)).float()

    def iteration(self) -> int:
        return int(self.it.item())


class GraphedTrainStep:
    """One training iteration of train.py:245-330 — differentiable render of a ray batch, loss, backward, optimiser step with
    the per-step learning-rate decay, coarse-table refresh — captured once as a hipGraph and replayed.

    An eager iteration queues ~150 launches (the library's kernels plus torch's fills, random numbers, loss arithmetic and
    autograd bookkeeping) from 1.3-2.0 ms of Python / ctypes time and leaves the device gaps between small kernels; a replay costs
    0.16 ms of host time and runs the same kernels 3 % faster (6.2 -> 6.03 ms per 8192-ray iteration, alternated on one box).

    The optimiser must be `FusedAdam(..., capturable=True, lr_factor=...)` (step count and lr schedule on the device).  The
    constructor runs `warmup` REAL iterations on the example batch (they train the model like any other iteration) and then
    captures one; `__call__(rays, target)` copies the batch into the graph's static inputs, replays, and returns the loss tensor
    of that iteration (overwritten by the next call).  Shapes are fixed: re-create the object after `upsample_volume_grid`
    (train.py:377-392 re-creates the optimiser there as well).  `loss_fn(rgb_map, target, alpha)` defaults to the MSE of
    train.py:250; regularisers go in there (egonerf_amd.losses).  `noise_fn` replaces torch.rand for the is_train jitter.

    Loss terms whose WEIGHTS change per iteration (train.py:295-309: the TV and ray-entropy weights decay by lr_factor per active
    iteration and are gated on iter_ignore_TV / iter_ignore_entropy) must not be Python floats - a float is frozen into the graph at
    capture.  A `loss_fn` whose fourth parameter is named `sched` / `schedule` (or any loss_fn with `schedule_aware=True`) receives `self.schedule` (a TrainSchedule: device-side iteration counter advanced
    inside the graph) and writes e.g. `sched.decayed(TV_weight_density, lr_factor, active_before=iter_ignore_TV) * model.TV_loss_density(tv)`;
    tests/test_hip_train_graph.py::test_graphed_step_with_decaying_regulariser_weights pins that against the eager loop."""

    def __init__(self, model, optimizer, rays, target, render_kwargs, loss_fn=None, warmup=3, noise_fn=None, start_iteration=0,
                 schedule_aware=None):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs FusedAdam(capturable=True): the step count and lr schedule must live on the device")
        self.model, self.opt, self.kw = model, optimizer, dict(render_kwargs)
        self.loss_fn = loss_fn or (lambda rgb, tgt, alpha: torch.mean((rgb - tgt) ** 2))
        # Explicit opt-in, not arity sniffing (a fourth parameter like reduction="mean" must not receive the schedule, a
        # *args loss must be able to): `schedule_aware=True`, or - when left None - a loss_fn parameter NAMED `sched` or `schedule`.
        if schedule_aware is None:
            import inspect
            try:
                schedule_aware = any(n in ("sched", "schedule") for n in inspect.signature(self.loss_fn).parameters)
            except (TypeError, ValueError):
                schedule_aware = False
        self._loss_takes_schedule = bool(schedule_aware)
        self.schedule = TrainSchedule(rays.device, start_iteration)
        # noise_fn(n_rays, n_samples, device) -> [n_rays, n_samples] in [0, 1): torch.rand by default; tests pin it
        self.noise_fn = noise_fn or (lambda n, m, dev: torch.rand(n, m, device=dev))
        self.rays, self.target = rays.detach().clone().float().contiguous(), target.detach().clone().float().contiguous()
        dev = self.rays.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):   # torch's capture recipe: warm up off the default stream
            for _ in range(max(int(warmup), 1)):
                self._iteration()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._iteration()
        self._params = [p for g in optimizer.param_groups for p in g["params"]]
        self.iterations = max(int(warmup), 1)  # the capture itself does not execute

    def _iteration(self):
        N = self.rays.shape[0]
        dev = self.rays.device
        kw = self.kw
        # the is_train noise comes from the device generator (graph-safe: every replay draws new numbers)
        jitter = self.noise_fn(N, kw["n_coarse"], dev)
        u = self.noise_fn(N, kw["n_fine"], dev) if kw.get("resampling") else None
        rgb, _depth, _bg, _env, alpha = self.model(self.rays, is_train=True, jitter=jitter, u=u, **kw)
        loss = self.loss_fn(rgb, self.target, alpha, self.schedule) if self._loss_takes_schedule else self.loss_fn(rgb, self.target, alpha)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        if kw.get("resampling"):
            self.model.update_coarse_sigma_grid()  # train.py:356-357
        self.schedule.it.add_(1.0)   # inside the graph: the next replay sees the next iteration index
        return loss.detach()

    def __call__(self, rays, target):
        self.rays.copy_(rays, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        self.graph.replay()
        self.iterations += 1
        # the replayed optimiser wrote the parameters through raw pointers: tell autograd and the model's packed-weight cache
        for p in self._params:
            torch.autograd.graph.increment_version(p)
        return self.loss
