"""Evaluation metrics of renderer.py:153-163 on device images: PSNR (renderer.py:156-157) and rgb_ssim (utils.py:104-152).
LPIPS needs external network weights and is out of scope.  The latitude-weighted ("WS") variants for equirectangular images
follow extra/ws_ssim.py:12-33 (marked TODO in renderer.py:89): row weights cos((i + 0.5 - N/2) pi / N), weighted mean of the
SSIM map; WS-PSNR applies the same weights to the squared error."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


@_lib.device_guard
def rgb_ssim(img0: torch.Tensor, img1: torch.Tensor, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, return_map=False):
    """utils.py:104-152 with the same signature; img0/img1 [H, W, 3] on the HIP device -> float (or the map tensor)."""
    assert img0.dim() == 3 and img0.shape[-1] == 3 and img0.shape == img1.shape
    if not (img0.is_cuda and img1.is_cuda):
        raise RuntimeError("rgb_ssim: needs HIP device tensors (the EgoNeRF path has no CPU fallback)")
    a, b = img0.detach().float().contiguous(), img1.detach().float().contiguous()
    H, W = a.shape[:2]
    Ho, Wo = H - filter_size + 1, W - filter_size + 1
    total = torch.zeros(1, dtype=torch.float64, device=a.device)
    smap = torch.empty(Ho, Wo, 3, device=a.device) if return_map else None
    _lib.check(_lib.load().ego_rgb_ssim(a.data_ptr(), b.data_ptr(), H, W, float(max_val), int(filter_size), float(filter_sigma),
                                        float(k1), float(k2), total.data_ptr(), _lib.ptr(smap), _lib.stream_handle()), "ego_rgb_ssim")
    return smap if return_map else float(total.item() / (Ho * Wo * 3))


def psnr(img: torch.Tensor, gt: torch.Tensor) -> float:
    """renderer.py:156-157."""
    loss = torch.mean((img - gt) ** 2)
    return float(-10.0 * np.log(loss.item()) / np.log(10.0))


def ws_weights(n_rows: int, row0: int = 0, n_total: int = None) -> np.ndarray:
    """extra/ws_ssim.py:12-14 (generate_ws) for rows row0 .. row0 + n_rows - 1 of an n_total-row panorama, float64:
    cos((i + 0.5 - N/2) pi / N).  ws_weights(N) == estws(map)[:, j] for an N-row map."""
    n_total = n_rows if n_total is None else n_total
    i = np.arange(row0, row0 + n_rows, dtype=np.float64)
    return np.cos((i + 0.5 - n_total / 2) * np.pi / n_total)


def weighted_map_mean(smap: torch.Tensor, row_weights) -> float:
    """extra/ws_ssim.py:29-31: sum(map * ws) / sum(ws) with ws constant along a row; smap [rows, cols] (any float dtype)."""
    w = torch.as_tensor(np.asarray(row_weights), dtype=torch.float64, device=smap.device)
    m = smap.to(torch.float64)
    return float(((m * w[:, None]).sum() / (w.sum() * m.shape[1])).item())


def ws_ssim(img0: torch.Tensor, img1: torch.Tensor, max_val=1.0, filter_size=11, filter_sigma=1.5):
    """(ssim, ws_ssim) of two [H, W, 3] equirectangular images on the HIP device.  The SSIM map is utils.py:104-152's
    ('valid' 11x11 Gaussian windows, so row i of the map is centred on image row i + filter_size // 2, which is the latitude
    its weight is taken at); channel mean, then extra/ws_ssim.py:29-31's weighted mean."""
    smap = rgb_ssim(img0, img1, max_val, filter_size, filter_sigma, return_map=True)  # [Ho, Wo, 3]
    m = smap.to(torch.float64).mean(-1)
    w = ws_weights(m.shape[0], filter_size // 2, img0.shape[0])
    return float(m.mean().item()), weighted_map_mean(m, w)


def ws_psnr(img: torch.Tensor, gt: torch.Tensor, max_val=1.0) -> float:
    """Latitude-weighted PSNR of [H, W, 3] equirectangular images: 10 log10(max^2 / (sum_ij w_i (x - y)^2 / (3 W sum_i w_i))),
    weights of extra/ws_ssim.py:12-14; reduces to renderer.py:156-157 for constant weights."""
    w = torch.as_tensor(ws_weights(img.shape[0]), dtype=torch.float64, device=img.device)
    d = img.to(torch.float64) - gt.to(img.device, torch.float64)
    wmse = ((d * d).sum((1, 2)) * w).sum() / (w.sum() * img.shape[1] * img.shape[2])
    return float(10.0 * np.log10(max_val ** 2 / wmse.item()))
