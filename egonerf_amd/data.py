"""Pose / ray ingestion for equirectangular datasets: host-side mirror of dataLoader/dataset_omniblender.py:11-95 (OmniBlender
`transform.json` + `{split}.txt` lists) with the rays generated on the HIP device (ego_erp_rays) instead of on the CPU.

Same attribute surface as the reference's dataset object where the render / training loop reads it: `poses [K,4,4]`, `img_wh`,
`near_far`, `center`, `scene_bbox [2,3]` (dataset_omniblender.py:22-32: camera-position centre +- (half diagonal of the camera
positions' extent + far)), `radius`, `all_rays`, `all_rgbs`, `image_paths`, `white_bg`, `indoor`.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Sequence

import numpy as np
import torch


class OmniBlenderDataset:
    def __init__(self, data_dir: str, split: str = "train", near_far: Sequence[float] = (0.1, 15.0), downsample: float = 1.0,
                 is_stack: bool = False, skip: int = 1, roi: Sequence[float] = (0, 1, 0, 1), device="cuda", load_images: bool = True,
                 **_ignored):
        self.root_dir, self.split, self.is_stack, self.skip = data_dir, split, is_stack, skip
        self.near_far, self.downsample, self.roi, self.device = list(near_far), downsample, list(roi), device
        self.white_bg = False
        self.blender2opencv = np.eye(4)                                            # dataset_interface.py:20
        self.img_wh = (int(2000 / downsample), int(1000 / downsample))              # dataset_omniblender.py:15
        self.img_list: List[str] = []
        self.image_paths: List[str] = []
        self._rays = self._rgbs = None
        self.read_meta(load_images)
        self.scene_bbox = self.get_scene_bbox()
        self.radius = (self.scene_bbox[1] - self.center).float().view(1, 1, 3)

    # dataset_omniblender.py:22-32
    def get_scene_bbox(self) -> torch.Tensor:
        cam = self.poses[:, :3, 3]
        self.center = cam.mean(0)
        trajectory_radius = (cam.max(0).values - cam.min(0).values).pow(2).sum(0).sqrt().div(2).float()
        return torch.stack([self.center - trajectory_radius - self.near_far[1], self.center + trajectory_radius + self.near_far[1]])

    # dataset_omniblender.py:34-95 (poses and image list; rays are produced by `rays()` on the device)
    def read_meta(self, load_images: bool) -> None:
        with open(os.path.join(self.root_dir, "transform.json"), "r") as f:
            self.meta = json.load(f)
        self.indoor = self.meta["indoor"]
        if self.split not in ("train", "test"):
            raise ValueError("Unknown split: {}".format(self.split))
        with open(os.path.join(self.root_dir, f"{self.split}.txt")) as f:
            self.img_list = [line.strip() for line in f if line.strip()]
        if self.split == "train":
            assert self.skip == 1, "skip must be 1 for training"
        self.img_list = self.img_list[::self.skip]
        names = [fr["file_path"].split(".")[0] for fr in self.meta["frames"]]
        poses, rgbs = [], []
        for img_name in self.img_list:
            frame = self.meta["frames"][names.index(img_name)]
            poses.append(torch.FloatTensor(np.array(frame["transform_matrix"]) @ self.blender2opencv))
            path = os.path.join(self.root_dir, "images", f"{frame['file_path']}")
            self.image_paths.append(path)
            if load_images and os.path.exists(path):
                rgbs.append(self._load_image(path))
        self.poses = torch.stack(poses)
        if rgbs:
            self._rgbs = rgbs

    def _load_image(self, path: str) -> torch.Tensor:
        """dataset_omniblender.py:66-77: PIL image (LANCZOS resize when downsampling) -> ToTensor -> [h*w, 3], alpha blended on white."""
        from PIL import Image
        img = Image.open(path)
        if self.downsample != 1.0:
            img = img.resize(self.img_wh, Image.LANCZOS)
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[..., None]
        t = torch.from_numpy(np.array(a)).float().div(255.0)                        # ToTensor: uint8 HWC -> float / 255
        t = t.view(-1, t.shape[-1])
        if t.shape[-1] == 4:
            t = t[:, :3] * t[:, -1:] + (1 - t[:, -1:])
        return t

    def rays(self, idx: int, device=None) -> torch.Tensor:
        """[h*w, 6] rays of image `idx`: get_ray_directions_360 + normalisation + get_rays(roi) (dataset_omniblender.py:41-43,79),
        generated on the device."""
        from .renderer import erp_rays
        w, h = self.img_wh
        h0, h1, w0, w1 = self.roi
        r0, r1, c0, c1 = int(h0 * h), int(h1 * h), int(w0 * w), int(w1 * w)       # ray_utils.py:100-103
        rays = erp_rays(h, w, self.poses[idx][:3].numpy(), device or self.device, r0, r1 - r0, normalize=True)
        if (c0, c1) != (0, w):
            rays = rays.view(r1 - r0, w, 6)[:, c0:c1].reshape(-1, 6)
        return rays

    @property
    def all_rays(self) -> torch.Tensor:
        """[K*h*w, 6] (or [K, h*w, 6] when is_stack), like dataset_omniblender.py:82-90; materialised on the device on first use."""
        if self._rays is None:
            per = [self.rays(i) for i in range(len(self.poses))]
            self._rays = torch.stack(per, 0) if self.is_stack else torch.cat(per, 0)
        return self._rays

    @property
    def all_rgbs(self) -> Optional[torch.Tensor]:
        if self._rgbs is None:
            return []
        if isinstance(self._rgbs, list):
            w, h = self.img_wh
            self._rgbs = torch.stack(self._rgbs, 0).reshape(-1, h, w, 3) if self.is_stack else torch.cat(self._rgbs, 0)
        return self._rgbs

    def world2ndc(self, points, lindisp=None):
        return (points - self.center.to(points.device)) / self.radius.to(points.device)

    def __len__(self):
        return len(self.poses)
