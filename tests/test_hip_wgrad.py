"""GPU: ego_weight_grad (G += A^T B over all rows, bias gradient from a ones column) against a float64 matmul: every
instantiation the training step uses, ragged row counts, tiny gradients (no fp16-style underflow), accumulation into G."""
import numpy as np
import pytest
import torch

from egonerf_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(A, ca, B, cb, ones_col, G=None):
    lib, st = _lib.load(), _lib.stream_handle()
    M = A.shape[0]
    if G is None:
        G = torch.zeros(32 * ((ca + 31) // 32), 160, device=DEV)
    _lib.check(lib.ego_weight_grad(A.data_ptr(), A.shape[1], ca, B.data_ptr(), B.shape[1], cb, ones_col, M, G.data_ptr(), 160, st),
               "ego_weight_grad")
    return G


@pytest.mark.parametrize("ca,lda,cb,ones_col", [(128, 128, 128, 128), (128, 128, 160, 154), (3, 3, 128, 128), (64, 64, 144, -1)])
@pytest.mark.parametrize("M", [1, 33, 4097, 70001])
def test_weight_grad_matches_float64(ca, lda, cb, ones_col, M):
    g = torch.Generator().manual_seed(ca * 7 + M)
    A = torch.randn(M, lda, generator=g) * torch.logspace(-9, -2, M).unsqueeze(1)   # gradients span 1e-9 .. 1e-2
    B = torch.randn(M, cb, generator=g).relu_()
    if 0 <= ones_col < cb:
        B[:, ones_col] = 0  # a padding column of the dump
    G = _run(A.to(DEV), ca, B.to(DEV), cb, ones_col).cpu().double()
    ref = A[:, :ca].double().T @ B.double()
    scale = float(ref.abs().max())
    cols = [c for c in range(cb) if c != ones_col]
    assert float((G[:ca, cols] - ref[:, cols]).abs().max()) <= 5e-5 * scale
    if ones_col >= 0:
        bias = A[:, :ca].double().sum(0)
        assert float((G[:ca, ones_col] - bias).abs().max()) <= 5e-5 * float(bias.abs().max())
    assert float(G[ca:].abs().max() if G.shape[0] > ca else 0.0) == 0.0  # padded rows stay zero


def test_weight_grad_accumulates_and_validates():
    A, B = torch.ones(40, 128, device=DEV), torch.ones(40, 128, device=DEV)
    G = _run(A, 128, B, 128, -1)
    G = _run(A, 128, B, 128, -1, G)
    assert float(G[:128, :128].min()) == 80.0 and float(G[:128, :128].max()) == 80.0
    lib = _lib.load()
    assert lib.ego_weight_grad(A.data_ptr(), 128, 129, B.data_ptr(), 128, 128, -1, 40, G.data_ptr(), 160, None) == -1
    assert b"weight_grad" in lib.ego_last_error()
    assert lib.ego_weight_grad(A.data_ptr(), 128, 96, B.data_ptr(), 128, 128, -1, 40, G.data_ptr(), 160, None) != 0  # no 3 x 4 instance
