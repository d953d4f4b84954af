"""The other appearance heads EgoNeRF.forward runs with (tensorBase.py:186-200): shadingMode 'MLP' (MLPRender,
tensorBase.py:107-129: MLPRender_Fea without the feature encoding, same state-dict keys) and 'RGB' (RGBRender, :37-39: colour = the three
appearance features, no sigmoid, no parameters), against tests/golden/heads.npz captured from the real reference
(oracle/capture_golden.py::capture_heads): the oracle's restatement on CPU, the HIP path on the GPU through the same Python surface."""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from tests.helpers import make_model, make_oracle

T = torch.from_numpy
HEADS = {   # keep in step with oracle/capture_golden.py::HEADS
    "mlp_head": dict(shadingMode="MLP", app_dim=27, view_pe=2, fea_pe=2, featureC=128),
    "mlp_head_small": dict(shadingMode="MLP", density_n_comp=(8, 8, 8), app_n_comp=(24, 24, 24), app_dim=12, view_pe=6, fea_pe=6, featureC=64),
    "rgb_head": dict(shadingMode="RGB", app_dim=3),
}
TRAIN_KW = dict(n_coarse=16, n_fine=16, resampling=True)


def _cfg(name):
    return synth.SceneConfig(n_voxel=20 ** 3, use_envmap=(name == "rgb_head"), envmap_res_H=16, **HEADS[name])


def _grad_keys(fx, name):
    pre = f"{name}/grad/"
    return [k[len(pre):] for k in fx.files if k.startswith(pre)]


@pytest.mark.parametrize("name", list(HEADS))
def test_oracle_reproduces_the_reference_heads(golden, name):
    fx = golden("heads")
    cfg = _cfg(name)
    sc = make_oracle(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))
    rays = T(synth.make_rays(48, seed=int(fx["seed_rays"])))
    q, dirs = T(fx[f"{name}/coords"]), T(fx[f"{name}/dirs"])
    af = sc.app_feature(q)
    assert af.shape[-1] == cfg.app_dim and float((af - T(fx[f"{name}/app"])).abs().max()) <= 2e-5
    assert float((sc.mlp_fea(dirs, T(fx[f"{name}/app"])) - T(fx[f"{name}/rgb_samples"])).abs().max()) <= 2e-6
    rgb, depth, _, _, alpha = sc.forward(rays, n_coarse=24)
    assert float((rgb - T(fx[f"{name}/nr_rgb"])).abs().max()) <= 2e-6 and float((alpha - T(fx[f"{name}/nr_alpha"])).abs().max()) <= 1e-5
    rgb, depth, *_ = sc.forward(rays, n_coarse=16, n_fine=16, resampling=True)
    assert float((rgb - T(fx[f"{name}/rs_rgb"])).abs().max()) <= 5e-6 and float((depth - T(fx[f"{name}/rs_depth"])).abs().max()) <= 5e-5


@pytest.mark.parametrize("name", list(HEADS))
def test_oracle_autograd_reproduces_the_reference_gradients_of_the_heads(golden, name):
    fx = golden("heads")
    cfg = _cfg(name)
    sc = make_oracle(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])))
    for v in sc.w.values():
        v.requires_grad_(True)
    sc.update_coarse_sigma_grid()
    rays = T(synth.make_rays(48, seed=int(fx["seed_rays"])))
    rgb = sc.forward(rays, is_train=True, jitter=T(fx[f"{name}/jitter"]), u=T(fx[f"{name}/u"]), **TRAIN_KW)[0]
    assert float((rgb.detach() - T(fx[f"{name}/train_rgb"])).abs().max()) <= 2e-6
    torch.mean((rgb - T(fx[f"{name}/gt"])) ** 2).backward()
    keys = _grad_keys(fx, name)
    assert ("renderModule.mlp.0.weight" in keys) == (name != "rgb_head")     # RGBRender has no parameters
    for k in keys:
        ref = fx[f"{name}/grad/{k}"]
        assert float((sc.w[k].grad - T(ref)).abs().max()) <= 5e-5 * max(float(np.abs(ref).max()), 1e-12), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(HEADS))
def test_hip_renders_the_other_heads_like_the_reference(golden, name):
    fx = golden("heads")
    cfg = _cfg(name)
    model = make_model(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])), "cuda")
    assert model.shadingMode == HEADS[name]["shadingMode"] and model.get_kwargs()["shadingMode"] == model.shadingMode
    # MLPRender is a module, RGBRender a plain function - as in the reference (tensorBase.py:37-39, EgoNeRF.py:152)
    assert getattr(model.renderModule, "__name__", type(model.renderModule).__name__) == {"MLP": "MLPRender", "RGB": "RGBRender"}[model.shadingMode]
    assert isinstance(model.renderModule, torch.nn.Module) == (name != "rgb_head")
    sd = model.state_dict()
    assert ("renderModule.mlp.0.weight" in sd) == (name != "rgb_head")
    rays = T(synth.make_rays(48, seed=int(fx["seed_rays"]))).cuda()
    q, dirs = T(fx[f"{name}/coords"]).cuda(), T(fx[f"{name}/dirs"]).cuda()
    with torch.no_grad():
        af = model.compute_appfeature(q)
        assert float((af.cpu() - T(fx[f"{name}/app"])).abs().max()) <= 2e-5
        rgb_s = model.renderModule(None, dirs, T(fx[f"{name}/app"]).cuda())
        assert float((rgb_s.cpu() - T(fx[f"{name}/rgb_samples"])).abs().max()) <= 1e-5
        rgb, depth, bg, env, alpha = model(rays, n_coarse=24, exp_sampling=True)
        assert float((rgb.cpu() - T(fx[f"{name}/nr_rgb"])).abs().max()) <= 1e-4      # north_star tolerance; measured ~1e-6
        assert float((alpha.cpu() - T(fx[f"{name}/nr_alpha"])).abs().max()) <= 1e-4
        assert float((depth.cpu() - T(fx[f"{name}/nr_depth"])).abs().max()) <= 1e-3
        rgb, depth, *_ = model(rays, n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)
        assert float((rgb.cpu() - T(fx[f"{name}/rs_rgb"])).abs().max()) <= 1e-4
        assert float((depth.cpu() - T(fx[f"{name}/rs_depth"])).abs().max()) <= 1e-3
        # ragged sizes + tile skip on an opaque field, against the oracle
        cfg2 = synth.SceneConfig(n_voxel=20 ** 3, density_shift=0.0, **HEADS[name])
        w2 = synth.make_weights(cfg2, seed=5)
        m2, o2 = make_model(cfg2, w2, "cuda"), make_oracle(cfg2, w2)
        r2 = T(synth.make_rays(333, seed=3))
        got = m2(r2.cuda(), n_coarse=37, exp_sampling=True)
        ref = o2.forward(r2, n_coarse=37)
        assert float((got[0].cpu() - ref[0]).abs().max()) <= 1e-4 and float((got[4].cpu() - ref[4]).abs().max()) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(HEADS))
def test_training_gradients_of_the_other_heads_vs_reference_autograd(golden, name):
    fx = golden("heads")
    cfg = _cfg(name)
    model = make_model(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])), "cuda")
    model.train()
    rays = T(synth.make_rays(48, seed=int(fx["seed_rays"]))).cuda()
    rgb, depth, _, _, alpha = model(rays, is_train=True, exp_sampling=True, use_coarse_sample=True, jitter=T(fx[f"{name}/jitter"]).cuda(),
                                    u=T(fx[f"{name}/u"]).cuda(), **TRAIN_KW)
    assert rgb.requires_grad and float((rgb.detach().cpu() - T(fx[f"{name}/train_rgb"])).abs().max()) <= 1e-4
    loss = torch.mean((rgb - T(fx[f"{name}/gt"]).cuda()) ** 2)
    assert abs(loss.item() - float(fx[f"{name}/loss"])) <= 1e-6
    loss.backward()
    named = dict(model.named_parameters())
    if cfg.use_envmap:
        named["envmap.emission"] = model.envmap.emission
    assert sorted(named) == sorted(_grad_keys(fx, name))
    for k, p in named.items():
        ref = fx[f"{name}/grad/{k}"]
        assert p.grad is not None, k
        g = p.grad.detach().cpu().numpy()
        assert g.shape == ref.shape, k
        # 2e-4 of the tensor's largest gradient against the reference's float32 autograd (as tests/test_model_shapes.py)
        assert float(np.abs(g - ref).max()) <= 2e-4 * max(float(np.abs(ref).max()), 1e-12), k
    from egonerf_amd.optim import FusedAdam
    FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99)).step()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mlp_head", "rgb_head"])
def test_other_heads_over_the_shipped_tables_take_the_sorted_scatter(name, monkeypatch):
    """Round 6: a head of another shape over 48-component appearance tables has ego_shade_backward_generic write `dv` in the tuned scatters'
    blocked layout (ldv = 0) and ego_scatter_app_sorted walk it - deterministic, and 13 x faster than the any-shape atomics at 8192 x 256.
    Same gradients as the row-major `dv` + ego_scatter_generic route (EGO_SORTED_WALK=0) up to the order of the float sums, the same bits
    twice, and the table gradients of a model whose tables are NOT 48-wide still arrive (through the atomics)."""
    from egonerf_amd import train
    cfg = synth.SceneConfig(n_voxel=24 ** 3, **HEADS[name])
    weights = synth.make_weights(cfg, seed=5)
    rays = T(synth.make_rays(300, seed=6)).cuda()
    g = torch.Generator().manual_seed(7)
    jit, u, gt = torch.rand(300, 16, generator=g).cuda(), torch.rand(300, 16, generator=g).cuda(), torch.rand(300, 3, generator=g).cuda()

    def grads():
        model = make_model(cfg, weights, "cuda")
        model.train()
        rgb = model(rays, is_train=True, exp_sampling=True, use_coarse_sample=True, jitter=jit, u=u, **TRAIN_KW)[0]
        torch.mean((rgb - gt) ** 2).backward()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    assert train._generic_head_sorted_app(make_model(cfg, weights, "cuda"), object(), 300 * 32)
    a, b = grads(), grads()
    for k in a:
        assert torch.equal(a[k], b[k]), ("sorted route, run to run", k)
    monkeypatch.setenv("EGO_SORTED_WALK", "0")
    assert not train._generic_head_sorted_app(make_model(cfg, weights, "cuda"), object(), 300 * 32)
    c = grads()
    assert sorted(a) == sorted(c)
    for k in a:
        scale = max(float(c[k].abs().max()), 1e-12)
        assert float((a[k] - c[k]).abs().max()) <= 2e-5 * scale, (k, float((a[k] - c[k]).abs().max()) / scale)


@pytest.mark.gpu
def test_blocked_dv_is_refused_for_other_table_widths():
    """ego_shade_backward_generic(ldv = 0) is the 48-component blocked layout: a 24-component model must be refused, not mis-indexed."""
    import ctypes as C
    from egonerf_amd import _lib
    cfg = _cfg("mlp_head_small")
    model = make_model(cfg, synth.make_weights(cfg, seed=5), "cuda")
    sc = model.scene(training=True)
    buf = torch.zeros(64 * 160, device="cuda")
    p = buf.data_ptr()
    rc = _lib.load().ego_shade_backward_generic(sc, p, p, p, p, 160, p, p, 160, p, p, p, p, 0, 2, 16, _lib.stream_handle())
    assert rc != 0 and b"blocked" in _lib.load().ego_last_error()
