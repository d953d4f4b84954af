"""Loss terms of the reference's training step that sit next to the render (train.py:245-330), as autograd nodes over the
HIP kernels of csrc/ego_reg.hip.  Each node computes its value AND the gradient in the same pass over the tables and hands
the stored gradient (scaled by the incoming one) back in backward, so `total_loss.backward()` reads exactly like train.py.

    utils.py:155-171   TVLoss                       -> TVLoss
    utils.py:175-183   ray_entropy_loss             -> ray_entropy_loss
    EgoNeRF.py:189-228 vector_comp_diffs, density_L1, TV_loss_density / TV_loss_app  (methods of model.EgoNeRF call in here)
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from . import _lib


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: needs a HIP device tensor (the EgoNeRF path has no CPU fallback)")


def _channel_last_ptr(p: torch.Tensor, what: str) -> int:
    """Tables are (1, C, H, W) tensors whose memory is [H][W][C] (model._channel_last_param)."""
    if p.dim() != 4 or p.shape[0] != 1 or not p.permute(0, 2, 3, 1).is_contiguous() or p.dtype != torch.float32:
        raise RuntimeError(f"{what}: expected a float32 (1, C, H, W) table with channel-last memory")
    return p.data_ptr()


class _TableReg(torch.autograd.Function):
    """value = sum over tables of one regulariser term; kind in {"tv", "l1", "ortho"}; scales[i] multiplies table i's term."""

    @staticmethod
    @_lib.device_guard
    def forward(ctx, kind: str, scales: Sequence[float], *tables: torch.Tensor):
        lib, st = _lib.load(), _lib.stream_handle()
        dev = tables[0].device
        value = torch.zeros(1, dtype=torch.float64, device=dev)
        need = [t.requires_grad for t in tables]
        grads: List[torch.Tensor] = []
        for t, s, nd in zip(tables, scales, need):
            _require_cuda(t, kind)
            ptr = _channel_last_ptr(t, kind)
            _, C_, H, W = t.shape
            g = torch.zeros_like(t) if nd else None  # zeros_like keeps the channel-last strides
            gp = None if g is None else g.data_ptr()
            if kind == "tv":
                _lib.check(lib.ego_tv_plane(ptr, C_, H, W, float(s), value.data_ptr(), gp, st), "ego_tv_plane")
            elif kind == "l1":
                _lib.check(lib.ego_l1_table(ptr, t.numel(), float(s), value.data_ptr(), gp, st), "ego_l1_table")
            elif kind == "ortho":
                if W != 1:
                    raise RuntimeError("ortho: expected a (1, C, n, 1) line table")
                _lib.check(lib.ego_line_ortho(ptr, C_, H, float(s), value.data_ptr(), gp, st), "ego_line_ortho")
            else:
                raise ValueError(kind)
            grads.append(g)
        ctx.grads = grads
        return value.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g_out):
        out = [None if g is None else g * g_out for g in ctx.grads]
        ctx.grads = None
        return (None, None, *out)


def table_regulariser(kind: str, tables: Sequence[torch.Tensor], scales: Sequence[float]) -> torch.Tensor:
    return _TableReg.apply(kind, list(scales), *tables)


class TVLoss(torch.nn.Module):
    """utils.py:155-171: TVLoss_weight * 2 * (sum dH^2 / count_h + sum dW^2 / count_w) / batch for one (1, C, H, W) plane."""

    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return table_regulariser("tv", [x], [float(self.TVLoss_weight)])


class _RayEntropy(torch.autograd.Function):
    @staticmethod
    @_lib.device_guard
    def forward(ctx, alpha: torch.Tensor):
        lib, st = _lib.load(), _lib.stream_handle()
        _require_cuda(alpha, "ray_entropy_loss")
        a = alpha.detach()
        if a.dtype != torch.float32 or a.dim() != 2 or not a.is_contiguous():
            a = a.float().contiguous()
        N, S = a.shape
        value = torch.zeros(1, dtype=torch.float64, device=a.device)
        g = torch.empty(N, S, device=a.device) if alpha.requires_grad else None
        _lib.check(lib.ego_ray_entropy(a.data_ptr(), N, S, S, value.data_ptr(), _lib.ptr(g), st), "ego_ray_entropy")
        ctx.g = g
        return value.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g_out):
        g = ctx.g
        ctx.g = None
        return None if g is None else g * g_out


def ray_entropy_loss(alpha: torch.Tensor) -> torch.Tensor:
    """utils.py:175-183: mean over rays of the entropy (bits) of alpha / (sum alpha + 1e-10); alpha [N, S(+1)] as returned
    by EgoNeRF.forward (the envmap's trailing ones column included, like the reference)."""
    return _RayEntropy.apply(alpha)
