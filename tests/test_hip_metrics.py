"""GPU: evaluation metrics (renderer.py:153-163) against the reference's own rgb_ssim / PSNR values (tests/golden/metrics.npz)
and the oracle's restatement at a full ERP image size."""
import numpy as np
import pytest
import torch

from egonerf_amd import metrics, synth
from egonerf_amd.renderer import erp_rays, evaluation
from tests.helpers import make_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_rgb_ssim_matches_reference(golden):
    fx = golden("metrics")
    a, b = torch.from_numpy(fx["img0"]).to(DEV), torch.from_numpy(fx["img1"]).to(DEV)
    assert abs(metrics.rgb_ssim(a, b, 1) - float(fx["ssim"])) <= 1e-9
    smap = metrics.rgb_ssim(a, b, 1, return_map=True)
    assert smap.shape == (30, 46, 3)
    assert float(np.abs(smap.cpu().numpy().astype(np.float64) - fx["ssim_map"]).max()) <= 1e-6  # the map is returned in float32
    assert abs(metrics.rgb_ssim(a, a, 1) - 1.0) <= 1e-12
    # constant image: its variance is pure rounding noise (~1e-17) that still bounds the covariance through the sqrt clip
    assert abs(metrics.rgb_ssim(torch.full_like(a, 0.25), b, 1) - float(fx["ssim_flat"])) <= 1e-7
    assert abs(metrics.rgb_ssim(a, b, 1, filter_size=7, filter_sigma=1.0) - float(fx["ssim_fs7"])) <= 1e-9
    assert abs(metrics.psnr(b, a) - float(fx["psnr"])) <= 1e-4
    with pytest.raises(RuntimeError, match="HIP device"):
        metrics.rgb_ssim(a.cpu(), b.cpu(), 1)
    with pytest.raises(RuntimeError, match="rgb_ssim"):
        metrics.rgb_ssim(a[:8], b[:8], 1)  # image smaller than the window


def test_rgb_ssim_ragged_size_vs_oracle():
    from oracle.egonerf_oracle import rgb_ssim as ref_ssim
    g = torch.Generator().manual_seed(5)
    for H, W in ((11, 11), (37, 53), (250, 333)):
        a = torch.rand(H, W, 3, generator=g)
        b = (a + 0.2 * torch.rand(H, W, 3, generator=g)).clamp(0, 1)
        got = metrics.rgb_ssim(a.to(DEV), b.to(DEV), 1)
        assert abs(got - ref_ssim(a.numpy(), b.numpy(), 1)) <= 1e-9, (H, W)


def test_evaluation_psnr_and_ssim_of_an_erp_render():
    """renderer.py:82-196 semantics on a small ERP view: gt = the same model's render + noise, so PSNR/SSIM have known values."""
    from oracle.egonerf_oracle import rgb_ssim as ref_ssim
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    model = make_model(cfg, synth.make_weights(cfg, seed=3), DEV)
    H, W = 24, 48
    rays = erp_rays(H, W, torch.eye(4)[:3], DEV)
    with torch.no_grad():
        img = model(rays, n_coarse=32, exp_sampling=True)[0].clamp(0, 1)
    noise = torch.from_numpy(synth.hash_uniform(4, 0, H * W * 3).reshape(H * W, 3).astype(np.float32)).to(DEV)
    gt = (img + 0.1 * (noise - 0.5)).clamp(0, 1)
    psnrs, ssims = evaluation([rays], [gt], (W, H), model, n_coarse=32, exp_sampling=True)
    mse = float(((img - gt) ** 2).mean())
    assert abs(psnrs[0] - (-10 * np.log10(mse))) <= 1e-3
    assert abs(ssims[0] - ref_ssim(img.view(H, W, 3).cpu().numpy(), gt.view(H, W, 3).cpu().numpy(), 1)) <= 1e-6
    assert 0.3 < ssims[0] < 1.0
    # the same call with the latitude-weighted variants (extra/ws_ssim.py weights)
    p2, s2, wp, wsim = evaluation([rays], [gt], (W, H), model, n_coarse=32, exp_sampling=True, ws_metrics=True)
    assert p2 == psnrs and abs(s2[0] - ssims[0]) <= 1e-7   # float32 SSIM map vs the kernel's own float64 sum
    assert abs(wp[0] - metrics.ws_psnr(img.view(H, W, 3), gt.view(H, W, 3))) <= 1e-9
    assert abs(wsim[0] - metrics.ws_ssim(img.view(H, W, 3), gt.view(H, W, 3))[1]) <= 1e-9 and 0.3 < wsim[0] < 1.0


def test_ws_metrics(golden):
    """WS-SSIM / WS-PSNR (extra/ws_ssim.py:12-33) on the device vs the reference-pinned oracle restatement."""
    from oracle.egonerf_oracle import rgb_ssim as ref_ssim, ws_psnr as ref_ws_psnr, ws_rows
    mx, fx = golden("metrics"), golden("ws_metrics")
    a, b = torch.from_numpy(mx["img0"]).to(DEV), torch.from_numpy(mx["img1"]).to(DEV)
    # the reference's weighting applied to the map's own rows (what capture_ws stored)
    m = metrics.rgb_ssim(a, b, 1, return_map=True).double().mean(-1)
    assert abs(metrics.weighted_map_mean(m, metrics.ws_weights(30)) - float(fx["wsssim"])) <= 1e-7   # float32 map
    ssim, wsssim = metrics.ws_ssim(a, b)
    assert abs(ssim - float(mx["ssim"])) <= 1e-7
    smap = ref_ssim(mx["img0"], mx["img1"], 1, return_map=True).mean(-1)
    w = ws_rows(40)[5:35]                                                      # map row i is centred on image row i + 5
    assert abs(wsssim - float((smap * w[:, None]).sum() / (w.sum() * smap.shape[1]))) <= 1e-7
    assert abs(metrics.ws_psnr(b, a) - ref_ws_psnr(mx["img1"], mx["img0"])) <= 1e-9
    # full ERP size: weights sum and a noise image with latitude-dependent error
    H, W = 1024, 2048
    g = torch.Generator().manual_seed(9)
    x = torch.rand(H, W, 3, generator=g)
    y = (x + 0.1 * torch.rand(H, W, 3, generator=g) * torch.linspace(0, 1, H)[:, None, None]).clamp(0, 1)
    assert abs(metrics.ws_psnr(y.to(DEV), x.to(DEV)) - ref_ws_psnr(y.numpy(), x.numpy())) <= 1e-8
