"""Training ray-index samplers: mirror of sampler.py:4-38 (CPU, numpy legacy RNG so that
np.random.seed(20221028) (train.py:413) reproduces the reference's index stream)."""
import numpy as np
import torch


class SimpleSampler:
    """Permutation epochs; re-permutes when fewer than two batches remain, so the tail of every
    permutation is dropped (sampler.py:11-16)."""

    def __init__(self, total, batch):
        self.total, self.batch = total, batch
        self.curr = total
        self.ids = None

    def nextids(self):
        self.curr += self.batch
        if self.curr + self.batch > self.total:
            self.ids = torch.LongTensor(np.random.permutation(self.total))
            self.curr = 0
        return self.ids[self.curr:self.curr + self.batch]


class ThetaImportanceSampler:
    """Uniform image and column, row drawn with p ~ cos(latitude)*lambda + 1 (sampler.py:19-38)."""

    def __init__(self, theta_importance_lambda, img_len, img_wh, batch, roi):
        self.img_len, self.batch = img_len, batch
        W, H = img_wh
        self.W = int(W * (roi[3] - roi[2]))
        self.H = int(H * (roi[1] - roi[0]))
        self.weight = self.get_weight(theta_importance_lambda, H, roi)

    def get_weight(self, theta_importance_lambda, h, roi):
        rows = np.arange(h)[int(h * roi[0]):int(h * roi[1])]
        lat = -(rows - h // 2) / h * np.pi
        w = np.cos(lat) * theta_importance_lambda + 1
        return w / np.sum(w)

    def nextids(self):
        img = np.random.choice(self.img_len, self.batch)
        col = np.random.choice(self.W, self.batch)
        row = np.random.choice(self.H, self.batch, p=self.weight)
        return img * self.W * self.H + (col + row * self.W)
