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

    def current_lrs(self) -> list:
        """The learning rates the NEXT step() applies, per group (train.py logs `param_group['lr']`; in capturable mode
        `group["lr"]` holds the base rate and the decay lives on the device).  A device synchronisation: for logging."""
        s = self.lr_scale() if self.capturable else 1.0
        return [float(g["lr"]) * s for g in self.param_groups]

    def steps_taken(self) -> int:
        """Optimiser steps executed so far, graph replays included (the host-side state['step'] does not advance during replays)."""
        if self.capturable and self._clock is not None:
            return int(self._clock[0])
        return max((int(st.get("step", 0)) for st in self.state.values()), default=0)

    def state_dict(self):
        """torch's state dict + the device clock of capturable mode (step count, lr scale): without it a reloaded optimiser would
        restart the bias correction and the learning-rate decay at t = 0.  state['step'] is synchronised from the clock."""
        if self.capturable and self._clock is not None:
            t = int(self._clock[0])
            for st in self.state.values():
                if "step" in st:
                    st["step"] = t
        sd = super().state_dict()
        if self.capturable and self._clock is not None:
            sd["ego_clock"] = [float(v) for v in self._clock.cpu()]
            sd["ego_lr_factor"] = self.lr_factor
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        clock = state_dict.pop("ego_clock", None)
        factor = state_dict.pop("ego_lr_factor", None)
        super().load_state_dict(state_dict)
        if clock is not None:
            dev = next(p for g in self.param_groups for p in g["params"]).device
            if self._clock is None:
                self._clock = torch.tensor(clock, dtype=torch.float64, device=dev)
            else:
                self._clock.copy_(torch.tensor(clock, dtype=torch.float64))   # in place: a captured graph keeps reading this buffer
            if factor is not None:
                self.lr_factor = float(factor)
        elif self.capturable:
            # a checkpoint written in non-capturable mode (no device clock): seed the clock from its step count, or the bias correction
            # and the decay would silently restart at t = 0.  Its param_groups[i]["lr"] are the rates the caller had already
            # decayed to (train.py:328-329), so they become the base rates and the device-side scale starts at 1.
            t = max((int(st.get("step", 0)) for st in self.state.values()), default=0)
            dev = next(p for g in self.param_groups for p in g["params"]).device
            seed = torch.tensor([float(t), 1.0, 0.0, 0.0], dtype=torch.float64, device=dev)
            if self._clock is None:
                self._clock = seed
            else:
                self._clock.copy_(seed)
            self._base_lrs = tuple(float(g["lr"]) for g in self.param_groups)

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
            base = tuple(float(g["lr"]) for g in self.param_groups)
            if getattr(self, "_base_lrs", base) != base:
                import warnings
                warnings.warn("FusedAdam(capturable=True): param_groups[i]['lr'] changed between steps; in this mode it is the BASE rate and the "
                              "per-step decay (lr_factor) is applied on the device - a caller that also multiplies group['lr'] decays twice, and a "
                              "captured graph keeps the rates of its capture", RuntimeWarning, stacklevel=2)
            self._base_lrs = base
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
