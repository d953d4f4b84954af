"""Multi-tensor Adam over `ego_adam_step` — a drop-in for the reference's `torch.optim.Adam(grad_vars, betas=(0.9, 0.99))`
(train.py:182,186): same parameter groups (model.get_optparam_groups), same update rule, and `param_groups[i]["lr"]` stays
writable so train.py:328-329's per-step exponential decay reads unchanged.  One launch updates all 33 tensors."""
from __future__ import annotations

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """`capturable=True` keeps the step count and the learning-rate schedule on the device (`ego_adam_step_graph`): every step()
    then uses `group["lr"] * scale` with `scale *= lr_factor` afterwards — train.py:328-329's per-step decay done by the kernel
    itself — so that a hipGraph capture of the whole training step (egonerf_amd.train.GraphedTrainStep) can be replayed.
    `group["lr"]` are then the BASE rates; `lr_scale()` reads the current scale back (a device synchronisation: for logging)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False, lr_factor=1.0):
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps <= 0:
            raise ValueError("FusedAdam: bad betas / eps")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.capturable, self.lr_factor = bool(capturable), float(lr_factor)
        self._clock = None  # device double[4]: step count, lr scale, two derived coefficients

    def lr_scale(self) -> float:
        return 1.0 if self._clock is None else float(self._clock[1])

    @torch.no_grad()
    @_lib.device_guard
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib, st = _lib.load(), _lib.stream_handle()
        batches = {}  # (beta1, beta2, eps, step) -> [AdamTensor]
        keep = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: needs HIP device parameters (the EgoNeRF path has no CPU fallback)")
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p)      # zeros_like keeps the parameter's (channel-last) strides
                    state["exp_avg_sq"] = torch.zeros_like(p)
                state["step"] += 1
                g = p.grad
                if g.stride() != p.stride() or g.dtype != torch.float32:
                    g = torch.empty_like(p).copy_(g)
                    keep.append(g)
                if p.dtype != torch.float32 or not p.permute(*sorted(range(p.dim()), key=lambda d: -p.stride(d))).is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be dense float32")
                t = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(), p.numel(),
                                    float(group["lr"]), 0)
                batches.setdefault((float(b1), float(b2), float(group["eps"]), state["step"]), []).append(t)
                torch.autograd.graph.increment_version(p)  # written through a raw pointer: keep autograd / caches honest
        if self.capturable and batches:
            if len({k[:3] for k in batches}) != 1:
                raise NotImplementedError("FusedAdam(capturable=True): one (betas, eps) setting for all groups (true for train.py:176-186)")
            ts = [t for group in batches.values() for t in group]
            (b1, b2, eps, _step) = next(iter(batches))
            if self._clock is None:
                dev = next(p for g in self.param_groups for p in g["params"]).device
                self._clock = torch.tensor([0.0, 1.0, 0.0, 0.0], dtype=torch.float64, device=dev)
            arr = (_lib.AdamTensor * len(ts))(*ts)
            _lib.check(lib.ego_adam_step_graph(arr, len(ts), b1, b2, eps, self.lr_factor, self._clock.data_ptr(), st), "ego_adam_step_graph")
            return loss
        for (b1, b2, eps, step), ts in batches.items():
            arr = (_lib.AdamTensor * len(ts))(*ts)
            _lib.check(lib.ego_adam_step(arr, len(ts), b1, b2, eps, step, st), "ego_adam_step")
        return loss
