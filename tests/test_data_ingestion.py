"""OmniBlender pose / image ingestion (egonerf_amd/data.py) against the reference's OmniBlenderDataset run on the same synthetic
three-frame dataset (tests/golden/omniblender.npz, rebuilt in a temp dir): host part on the CPU, ray generation on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from egonerf_amd.data import OmniBlenderDataset


def _build(fx, d):
    from PIL import Image
    os.makedirs(os.path.join(d, "images"))
    open(os.path.join(d, "transform.json"), "w").write(str(fx["frames_json"]))
    open(os.path.join(d, "train.txt"), "w").write(str(fx["train_list"]))
    open(os.path.join(d, "test.txt"), "w").write(str(fx["test_list"]))
    for k in fx.files:
        if k.startswith("png/"):
            Image.fromarray(fx[k], "RGBA").save(os.path.join(d, "images", k[4:] + ".png"))
    return str(d)


@pytest.mark.parametrize("split,stack", [("train", False), ("test", True)])
def test_poses_bbox_and_images_match_reference(golden, tmp_path, split, stack):
    fx = golden("omniblender")
    ds = OmniBlenderDataset(_build(fx, tmp_path), split=split, near_far=[0.01, 15.0], downsample=250.0, is_stack=stack, device="cpu")
    assert tuple(ds.img_wh) == tuple(fx[f"{split}/img_wh"]) == (8, 4) and len(ds) == fx[f"{split}/poses"].shape[0]
    assert np.array_equal(ds.poses.numpy(), fx[f"{split}/poses"])
    assert np.array_equal(ds.center.numpy(), fx[f"{split}/center"]) and np.array_equal(ds.scene_bbox.numpy(), fx[f"{split}/scene_bbox"])
    assert np.array_equal(ds.radius.numpy(), fx[f"{split}/radius"])
    assert np.array_equal(ds.all_rgbs.numpy(), fx[f"{split}/all_rgbs"])   # RGBA blended on white, uint8 / 255
    assert ds.indoor is True and ds.white_bg is False
    with pytest.raises(ValueError):
        OmniBlenderDataset(str(tmp_path), split="val", device="cpu")


@pytest.mark.gpu
def test_rays_on_device_match_reference(golden, tmp_path):
    fx = golden("omniblender")
    d = _build(fx, tmp_path)
    ds = OmniBlenderDataset(d, split="train", near_far=[0.01, 15.0], downsample=250.0, device="cuda")
    assert float((ds.all_rays.cpu() - torch.from_numpy(fx["train/all_rays"])).abs().max()) <= 1e-6
    st = OmniBlenderDataset(d, split="test", near_far=[0.01, 15.0], downsample=250.0, is_stack=True, device="cuda")
    assert st.all_rays.shape == (1, 32, 6) and float((st.all_rays.cpu() - torch.from_numpy(fx["test/all_rays"])).abs().max()) <= 1e-6
    roi = OmniBlenderDataset(d, split="train", near_far=[0.01, 15.0], downsample=250.0, roi=[0.25, 1.0, 0.0, 0.5], device="cuda")
    assert float((roi.all_rays.cpu() - torch.from_numpy(fx["roi/all_rays"])).abs().max()) <= 1e-6
