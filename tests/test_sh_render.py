"""SHRender (models/tensorBase.py:30-34; the "SH eval" of the hot path's alternative colour head) against values captured
from the reference's own function (tests/golden/sh_render.npz)."""
import numpy as np
import pytest
import torch


def test_oracle_sh_render(golden):
    from oracle.egonerf_oracle import sh_render
    fx = golden("sh_render")
    got = sh_render(torch.from_numpy(fx["dirs"]), torch.from_numpy(fx["features"]))
    assert float((got - torch.from_numpy(fx["rgb"])).abs().max()) <= 5e-7


@pytest.mark.gpu
def test_hip_sh_render(golden):
    from egonerf_amd.model import SHRender
    fx = golden("sh_render")
    d, f = torch.from_numpy(fx["dirs"]).cuda(), torch.from_numpy(fx["features"]).cuda()
    got = SHRender(None, d, f)
    assert got.shape == (300, 3) and float((got.cpu() - torch.from_numpy(fx["rgb"])).abs().max()) <= 1e-6
    assert float(got.min()) >= 0.0 and bool((got == 0).any())  # the relu is active on this input
    assert SHRender(None, d[:0], f[:0]).shape == (0, 3)
    with pytest.raises(RuntimeError, match="HIP device"):
        SHRender(None, d.cpu(), f.cpu())
