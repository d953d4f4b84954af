"""exp_sampling = False (TensorBase.sample_ray, tensorBase.py:308-327: aabb entry clamped to [near, far] + uniform stepSize
steps) feeding EgoNeRF.forward, against tests/golden/tiny_uniform.npz captured from the reference: the oracle's restatement on
CPU, the HIP path (explicit first-pass distances through ego_render_args.z_coarse / ego_march_density z_in) on the GPU."""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from tests.helpers import make_model, make_oracle

T = torch.from_numpy


def _cfg():
    return synth.SceneConfig(n_voxel=20 ** 3)


def test_oracle_uniform_sampling(golden):
    fx = golden("tiny_uniform")
    cfg = _cfg()
    sc = make_oracle(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))
    rays = T(fx["rays"])
    xyz, z = sc.sample_ray(rays[:, :3], rays[:, 3:6], 24)
    assert np.array_equal(z.numpy(), fx["z_eval"]) and np.array_equal(xyz.numpy(), fx["xyz_eval"])
    assert abs(float(z[0, 1] - z[0, 0]) - float(fx["step_size"])) <= 1e-6
    rgb, depth, _, _, alpha = sc.forward(rays, n_coarse=24, exp_sampling=False)
    assert float((rgb - T(fx["nr_rgb"])).abs().max()) <= 2e-6 and float((alpha - T(fx["nr_alpha"])).abs().max()) <= 5e-6  # ATen CPU paths differ by box
    assert float((depth - T(fx["nr_depth"])).abs().max()) <= 2e-5
    rgb, depth, *_ = sc.forward(rays, n_coarse=16, n_fine=16, resampling=True, exp_sampling=False)
    assert float((rgb - T(fx["rs_rgb"])).abs().max()) <= 1e-6 and float((depth - T(fx["rs_depth"])).abs().max()) <= 2e-5
    rgb, depth, *_ = sc.forward(rays, n_coarse=16, n_fine=16, resampling=True, exp_sampling=False, is_train=True,
                                jitter=T(fx["tr_jitter"]), u=T(fx["tr_u"]))
    assert float((rgb - T(fx["tr_rgb"])).abs().max()) <= 1e-6 and float((depth - T(fx["tr_depth"])).abs().max()) <= 2e-5


def test_host_uniform_distances_reproduce_the_reference(golden):
    """The product's host-side schedule (EgoNeRF.sample_ray_z) on CPU tensors — no kernel involved."""
    fx = golden("tiny_uniform")
    cfg = _cfg()
    model = make_model(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])), "cpu")
    assert abs(float(model.stepSize) - float(fx["step_size"])) <= 1e-6
    z = model.sample_ray_z(T(fx["rays"]), 24)
    assert np.array_equal(z.numpy(), fx["z_eval"])
    xyz, z, inside = model.sample_ray(T(fx["rays"])[:, :3], T(fx["rays"])[:, 3:6], is_train=False, N_samples=24)
    assert np.array_equal(xyz.numpy(), fx["xyz_eval"]) and inside.shape == (64, 24)
    zj = model.sample_ray_z(T(fx["rays"]), 16, T(fx["tr_jitter"]))
    ref = zj[:, :1] - float(model.stepSize) * T(fx["tr_jitter"])[:, :1] + float(model.stepSize) * (torch.arange(16)[None] + T(fx["tr_jitter"]))
    assert float((zj - ref).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_hip_uniform_sampling(golden):
    fx = golden("tiny_uniform")
    cfg = _cfg()
    model = make_model(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])), "cuda")
    rays = T(fx["rays"]).cuda()
    with torch.no_grad():
        rgb, depth, _, _, alpha = model(rays, n_coarse=24, exp_sampling=False)
        assert float((rgb.cpu() - T(fx["nr_rgb"])).abs().max()) <= 1e-4
        assert float((alpha.cpu() - T(fx["nr_alpha"])).abs().max()) <= 1e-4
        assert float((depth.cpu() - T(fx["nr_depth"])).abs().max()) <= 1e-3
        rgb, depth, *_ = model(rays, n_coarse=16, n_fine=16, exp_sampling=False, resampling=True)
        assert float((rgb.cpu() - T(fx["rs_rgb"])).abs().max()) <= 1e-4
        assert float((depth.cpu() - T(fx["rs_depth"])).abs().max()) <= 1e-3
        rgb, depth, *_ = model(rays, n_coarse=16, n_fine=16, exp_sampling=False, resampling=True, is_train=True,
                               jitter=T(fx["tr_jitter"]).cuda(), u=T(fx["tr_u"]).cuda())
        assert float((rgb.cpu() - T(fx["tr_rgb"])).abs().max()) <= 1e-4
        assert float((depth.cpu() - T(fx["tr_depth"])).abs().max()) <= 1e-3


def test_oracle_uniform_sampling_mixed_entry_distances(golden):
    """Rays starting outside the aabb enter it at different distances; in eval the reference samples per ray but measures every ray
    with ray 0's distances (EgoNeRF.py:515-516).  Captured from the reference (tiny_uniform_mixed.npz); the oracle restates it."""
    fx = golden("tiny_uniform_mixed")
    cfg = _cfg()
    sc = make_oracle(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))
    rays = T(fx["rays"])
    _xyz, z = sc.sample_ray(rays[:, :3], rays[:, 3:6], 24)
    assert np.array_equal(z.numpy(), fx["z_eval"]) and len(np.unique(fx["z_eval"][:, 0])) > 20
    rgb, depth, _, _, alpha = sc.forward(rays, n_coarse=24, exp_sampling=False)
    assert float((rgb - T(fx["nr_rgb"])).abs().max()) <= 2e-6 and float((alpha - T(fx["nr_alpha"])).abs().max()) <= 5e-6
    assert float((depth - T(fx["nr_depth"])).abs().max()) <= 2e-5
    rgb, depth, *_ = sc.forward(rays, n_coarse=16, n_fine=16, resampling=True, exp_sampling=False)
    assert float((rgb - T(fx["rs_rgb"])).abs().max()) <= 2e-6 and float((depth - T(fx["rs_depth"])).abs().max()) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("envmap", [False, True])
def test_hip_uniform_sampling_mixed_entry_distances(golden, envmap):
    """the same case on the GPU (it used to raise NotImplementedError): reference goldens without an envmap,
    the oracle with one (alpha gets its extra column, bg / env maps appear)."""
    fx = golden("tiny_uniform_mixed")
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=envmap, envmap_res_H=16)
    w = synth.make_weights(cfg, seed=int(fx["seed_weights"]))
    model = make_model(cfg, w, "cuda")
    rays = T(fx["rays"])
    z = model.sample_ray_z(rays.cuda(), 24)
    assert np.array_equal(z.cpu().numpy(), fx["z_eval"])
    with torch.no_grad():
        for kw in (dict(n_coarse=24), dict(n_coarse=16, n_fine=16, resampling=True)):
            got = model(rays.cuda(), exp_sampling=False, **kw)
            if envmap:
                ref = make_oracle(cfg, w).forward(rays, exp_sampling=False, **kw)
                assert got[4].shape == ref[4].shape and float((got[3].cpu() - ref[3]).abs().max()) <= 1e-5
                assert float((got[2].cpu() - ref[2]).abs().max()) <= 1e-4
            else:
                tag = "rs" if kw.get("resampling") else "nr"
                ref = (T(fx[f"{tag}_rgb"]), T(fx[f"{tag}_depth"]), None, None, T(fx[f"{tag}_alpha"]))
            assert float((got[0].cpu() - ref[0]).abs().max()) <= 1e-4
            assert float((got[1].cpu() - ref[1]).abs().max()) <= 1e-3 * 15.0
            assert float((got[4].cpu() - ref[4]).abs().max()) <= 1e-4
            none = model(rays.cuda(), exp_sampling=False, need_alpha=False, **kw)
            assert none[4] is None and torch.equal(none[0], got[0])


@pytest.mark.gpu
def test_hip_uniform_sampling_training_gradients_vs_oracle_autograd(golden):
    """The differentiable path with explicit first-pass distances: table / MLP gradients against float64 autograd through the
    oracle on the same pinned noise."""
    fx = golden("tiny_uniform")
    cfg = _cfg()
    w = synth.make_weights(cfg, seed=int(fx["seed_weights"]))
    model = make_model(cfg, w, "cuda")
    model.train()
    rays = T(fx["rays"]).cuda()
    gt = T(synth.hash_uniform(21, 0, 64 * 3).reshape(64, 3).astype(np.float32))
    rgb, *_ = model(rays, n_coarse=16, n_fine=16, exp_sampling=False, resampling=True, is_train=True,
                    jitter=T(fx["tr_jitter"]).cuda(), u=T(fx["tr_u"]).cuda())
    assert rgb.requires_grad
    assert float((rgb.detach().cpu() - T(fx["tr_rgb"])).abs().max()) <= 1e-4
    torch.mean((rgb - gt.cuda()) ** 2).backward()

    oracle = make_oracle(cfg, w)
    for v in oracle.w.values():
        v.requires_grad_(True)
    ref, *_ = oracle.forward(T(fx["rays"]), n_coarse=16, n_fine=16, resampling=True, exp_sampling=False, is_train=True,
                             jitter=T(fx["tr_jitter"]), u=T(fx["tr_u"]))
    torch.mean((ref - gt) ** 2).backward()
    for k, p in model.named_parameters():
        r = oracle.w[k].grad
        r = torch.zeros_like(oracle.w[k]) if r is None else r
        scale = max(float(r.abs().max()), 1e-12)
        assert float((p.grad.detach().cpu() - r).abs().max()) / scale <= 3e-4, k
