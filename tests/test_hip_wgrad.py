"""GPU: ego_weight_grad (G += A^T B over all rows, bias gradient from a ones column) against a float64 matmul: every
instantiation the training step uses, ragged row counts, tiny gradients (no fp16-style underflow), accumulation into G."""
import numpy as np
import pytest
import torch

from egonerf_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _blocked(X):
    """Row-major [M, W] -> the shade kernels' dump layout [tile][quad pair q][lane = 32 h + j][4] (logical column 8 q + 4 h + c)."""
    M, W = X.shape
    Mp = (M + 31) // 32 * 32
    P = torch.full((Mp, W), float("nan"))   # padding rows must never be read
    P[:M] = X
    return P.view(Mp // 32, 32, W // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().view(Mp, W)


def _scaled_half(X):
    """Row-major fp32 [M, 128] -> ego_shade_backward's scaled-fp16 layout [tile][k-step s][lane = 32 h + j][8 halves] (element e =
    logical column 8 (2 s + e / 4) + 4 h + e % 4) + the per-row power of two that puts the row's largest magnitude into
    [2^12, 2^13), and the fp32 matrix those halves stand for."""
    M, W = X.shape
    assert W == 128
    Mp = (M + 31) // 32 * 32
    amax = X.abs().amax(1).clamp_min(1e-38)
    k = 12 - torch.floor(torch.log2(amax))
    scale, inv = torch.exp2(k), torch.exp2(-k)
    H = (X * scale[:, None]).half()
    exact = H.float() * inv[:, None]
    P = torch.full((Mp, W), float("nan"), dtype=torch.float16)   # padding rows must never be read
    P[:M] = H
    # column = 8 (2 s + e4) + 4 h + c  ->  [tile][j][s][e4][h][c] -> [tile][s][h][j][e4][c]
    P = P.view(Mp // 32, 32, 8, 2, 2, 4).permute(0, 2, 4, 1, 3, 5).contiguous().view(Mp, W)
    return P, inv, exact


def _half_blocked(X):
    """Row-major fp32 [M, W] (W = 128 or 160) -> the forward's x / h1 / h2 dump layout: halves [tile][k-step s][lane = 32 h + j][8]
    (element e = logical column 8 (2 s + e / 4) + 4 h + e % 4), and the fp32 matrix those halves stand for."""
    M, W = X.shape
    Mp = (M + 31) // 32 * 32
    H = X.half()
    P = torch.full((Mp, W), float("nan"), dtype=torch.float16)   # padding rows must never be read
    P[:M] = H
    P = P.view(Mp // 32, 32, W // 16, 2, 2, 4).permute(0, 2, 4, 1, 3, 5).contiguous().view(Mp, W)
    return P, H.float()


def _run(A, ca, B, cb, ones_col, G=None, a_blocked=0, b_blocked=0, M=None, a_scale=None):
    lib, st = _lib.load(), _lib.stream_handle()
    M = A.shape[0] if M is None else M
    if G is None:
        G = torch.zeros(32 * ((ca + 31) // 32), 160, device=DEV)
    _lib.check(lib.ego_weight_grad(A.data_ptr(), A.shape[1], ca, a_blocked, _lib.ptr(a_scale), B.data_ptr(), B.shape[1], cb, b_blocked,
                                   ones_col, M, G.data_ptr(), 160, st), "ego_weight_grad")
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
    # the same product with the operands in the tile-blocked dump layout (the forms the training step uses)
    if cb in (128, 160, 144):
        ab = lda == 128
        Gb = _run((_blocked(A) if ab else A).to(DEV), ca, _blocked(B).to(DEV), cb, ones_col, a_blocked=int(ab), b_blocked=1, M=M).cpu().double()
        assert float((Gb[:ca, cols] - ref[:, cols]).abs().max()) <= 5e-5 * scale
        assert torch.isfinite(Gb).all()
    # ... and with A as the shade backward writes it: scaled fp16 + one power of two per row.  Against the product of the values
    # the halves stand for, the pass is as exact as the fp32 forms; against the unrounded A the fp16 rounding (2^-12 per element,
    # unbiased) shows, averaged over the rows
    if lda == 128 and cb in (128, 160):
        Ah, inv, exact = _scaled_half(A)
        Gh = _run(Ah.to(DEV), ca, _blocked(B).to(DEV), cb, ones_col, a_blocked=2, b_blocked=1, M=M, a_scale=inv.to(DEV)).cpu().double()
        ref_h = exact.double().T @ B.double()
        assert float((Gh[:ca, cols] - ref_h[:, cols]).abs().max()) <= 5e-5 * scale
        assert float((Gh[:ca, cols] - ref[:, cols]).abs().max()) <= 3e-4 * scale
        assert torch.isfinite(Gh).all()
        if ones_col >= 0:
            assert float((Gh[:ca, ones_col] - exact.double().sum(0)).abs().max()) <= 5e-5 * float(bias.abs().max())
        # ... and with B as the training forward writes x / h1 / h2: halves in the same operand order (b_layout 2)
        Bh, Bexact = _half_blocked(B)
        Ghh = _run(Ah.to(DEV), ca, Bh.to(DEV), cb, ones_col, a_blocked=2, b_blocked=2, M=M, a_scale=inv.to(DEV)).cpu().double()
        ref_hh = exact.double().T @ Bexact.double()
        assert float((Ghh[:ca, cols] - ref_hh[:, cols]).abs().max()) <= 5e-5 * scale
        assert float((Ghh[:ca, cols] - ref[:, cols]).abs().max()) <= (4e-4 if M > 1000 else 1e-3) * scale   # two operands of 2^-12 each, averaged over the rows
        assert torch.isfinite(Ghh).all()
        if ones_col >= 0:
            assert float((Ghh[:ca, ones_col] - exact.double().sum(0)).abs().max()) <= 5e-5 * float(bias.abs().max())
    if ca == 3 and cb == 128:   # d(W3) = do^T h2: ragged row-major A, halves B
        Bh, Bexact = _half_blocked(B)
        G3 = _run(A.to(DEV), ca, Bh.to(DEV), cb, ones_col, a_blocked=0, b_blocked=2, M=M).cpu().double()
        assert float((G3[:ca, cols] - A[:, :ca].double().T @ Bexact.double()[:, cols]).abs().max()) <= 5e-5 * scale
        assert float((G3[:ca, ones_col] - bias).abs().max()) <= 5e-5 * float(bias.abs().max())


@pytest.mark.parametrize("M", [1, 33, 4097, 70001])
def test_weight_grad_grid_routed_layout(M):
    """a_layout 3 (ego_shade_backward's dfe): A is [M][32], row m stands for 64 logical columns with its 32 values in the block
    of the sample's grid (coords[m][3] != 0 -> columns 32..63) and zeros in the other."""
    g = torch.Generator().manual_seed(M)
    A32 = torch.randn(M, 32, generator=g) * torch.logspace(-9, -2, M).unsqueeze(1)
    B = torch.randn(M, 144, generator=g)
    grid = (torch.rand(M, generator=g) > 0.4)
    coords = torch.rand(M, 4, generator=g)
    coords[:, 3] = grid.float()
    A64 = torch.zeros(M, 64)
    A64[~grid, :32] = A32[~grid]
    A64[grid, 32:] = A32[grid]
    ref = A64.double().T @ B.double()
    G = _run(A32.to(DEV), 64, _blocked(B).to(DEV), 144, -1, a_blocked=3, b_blocked=1, M=M, a_scale=coords.to(DEV)).cpu().double()
    assert float((G[:64, :144] - ref).abs().max()) <= 5e-5 * float(ref.abs().max())
    # ... and equals the plain row-major form on the expanded matrix
    G0 = _run(A64.to(DEV), 64, _blocked(B).to(DEV), 144, -1, a_blocked=0, b_blocked=1, M=M).cpu().double()
    assert float((G - G0).abs().max()) <= 3e-6 * float(ref.abs().max())   # both runs add their workgroups' partial products with float atomics (order varies: 1.04e-6 seen)
    # ... and with B as the training forward writes v: halves in the kernels' operand order (b_layout 2)
    Bh, Bexact = _half_blocked(B)
    Gh = _run(A32.to(DEV), 64, Bh.to(DEV), 144, -1, a_blocked=3, b_blocked=2, M=M, a_scale=coords.to(DEV)).cpu().double()
    assert float((Gh[:64, :144] - A64.double().T @ Bexact.double()).abs().max()) <= 5e-5 * float(ref.abs().max())


def test_weight_grad_accumulates_and_validates():
    A, B = torch.ones(40, 128, device=DEV), torch.ones(40, 128, device=DEV)
    G = _run(A, 128, B, 128, -1)
    G = _run(A, 128, B, 128, -1, G)
    assert float(G[:128, :128].min()) == 80.0 and float(G[:128, :128].max()) == 80.0
    lib = _lib.load()
    assert lib.ego_weight_grad(A.data_ptr(), 128, 129, 0, None, B.data_ptr(), 128, 128, 0, -1, 40, G.data_ptr(), 160, None) == -1
    assert b"weight_grad" in lib.ego_last_error()
    assert lib.ego_weight_grad(A.data_ptr(), 128, 96, 0, None, B.data_ptr(), 128, 128, 0, -1, 40, G.data_ptr(), 160, None) != 0  # no 3 x 4 instance
    assert lib.ego_weight_grad(A.data_ptr(), 128, 128, 2, None, B.data_ptr(), 128, 128, 1, -1, 40, G.data_ptr(), 160, None) == -1  # no a_scale


@pytest.mark.parametrize("M", [2048, 70001])
def test_both_halves_form_with_wild_scales_and_zero_rows(M):
    """k_wgrad_h takes the largest per-row scale of a 32-row step as its reference: neighbouring rows whose gradients differ by seven
    orders of magnitude (consecutive samples of a ray do) and rows that are all zero (ego_shade_backward writes scale 0 for them - a 1.0
    standing for "nothing" would flush the step) must still give the product of the values the halves stand for."""
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, 128, generator=g) * torch.pow(10.0, -9 + 7 * torch.rand(M, generator=g)).unsqueeze(1)
    dead = torch.rand(M, generator=g) < 0.15
    A[dead] = 0
    B = torch.randn(M, 160, generator=g)
    B[:, 154] = 0
    Ah, inv, exact = _scaled_half(torch.where(dead[:, None], torch.ones_like(A), A))
    Mp = Ah.shape[0]
    # zero rows: payload 0, scale 0 (the layout permutes rows inside a tile: rebuild the payload from the masked matrix)
    Az = torch.where(dead[:, None], torch.zeros_like(A), A)
    amax = Az.abs().amax(1).clamp_min(1e-38)
    k = 12 - torch.floor(torch.log2(amax))
    scale, inv = torch.exp2(k), torch.exp2(-k)
    scale[dead], inv[dead] = 0.0, 0.0
    H = (Az * scale[:, None]).half()
    exact = H.float() * inv[:, None]
    P = torch.zeros(Mp, 128, dtype=torch.float16)
    P[:M] = H
    Ah = P.view(Mp // 32, 32, 8, 2, 2, 4).permute(0, 2, 4, 1, 3, 5).contiguous().view(Mp, 128)
    Bh, Bexact = _half_blocked(B)
    G = _run(Ah.to(DEV), 128, Bh.to(DEV), 160, 154, a_blocked=2, b_blocked=2, M=M, a_scale=inv.to(DEV)).cpu().double()
    ref = exact.double().T @ Bexact.double()
    cols = [c for c in range(160) if c != 154]
    scale_ref = float(ref.abs().max())
    assert float((G[:128, cols] - ref[:, cols]).abs().max()) <= 5e-5 * scale_ref
    assert float((G[:128, 154] - exact.double().sum(0)).abs().max()) <= 5e-5 * float(exact.double().sum(0).abs().max())
