"""Shared test plumbing: build the product model / the oracle from one synthetic weight set."""
import contextlib
import io

import numpy as np
import torch

from egonerf_amd import synth


def make_coords(cfg, device):
    return synth.build_coords(cfg, device)


def make_model(cfg, weights, device="cuda"):
    """egonerf_amd EgoNeRF with the reference's ctor kwargs (train.py:163-171 resolved values)."""
    return synth.build_model(cfg, weights, device)


def make_oracle(cfg, weights, dtype=torch.float32):
    from oracle.egonerf_oracle import OracleScene
    return OracleScene(cfg, weights, dtype=dtype)


def maxerr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
