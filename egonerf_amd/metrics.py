"""Evaluation metrics of renderer.py:153-163 on device images: PSNR (renderer.py:156-157) and rgb_ssim (utils.py:104-152).
LPIPS needs external network weights and is out of scope; extra/ws_ssim.py depends on torchmetrics (absent in the reference's
own requirements) and is marked TODO in renderer.py:89, so it is not mirrored."""
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
