"""GPU: wrong trailing dimensions raise IndexError (what the reference's `[..., k]` indexing does) instead of being silently
reinterpreted by the raw-pointer calls — e.g. train.py:268-270's sparsity term hands compute_densityfeature [N, 3] points."""
import pytest
import torch

from egonerf_amd import synth
from tests.helpers import make_model

pytestmark = pytest.mark.gpu


def test_wrong_shapes_raise():
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    model = make_model(cfg, synth.make_weights(cfg, seed=1), "cuda")
    z3, z7 = torch.zeros(5, 3, device="cuda"), torch.zeros(5, 7, device="cuda")
    for call in (lambda: model.compute_densityfeature(z3), lambda: model.compute_coarse_densityfeature(z3),
                 lambda: model.compute_appfeature(z3), lambda: model.coordinates.normalize_coord(z3),
                 lambda: model.coordinates.from_cartesian(z7), lambda: model(z3, n_coarse=8, exp_sampling=True),
                 lambda: model(torch.zeros(6, device="cuda"), n_coarse=8, exp_sampling=True)):
        with pytest.raises(IndexError):
            call()
    assert model.compute_densityfeature(z7).shape == (5,) and model.compute_appfeature(z7[None]).shape == (1, 5, 27)
