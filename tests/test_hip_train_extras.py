"""GPU: the rest of the reference's training step (train.py:245-330) against tests/golden/train_extras.npz (captured from the
real reference's autograd by oracle/capture_golden.py): ray-entropy gradient through `alpha`, envmap-emission gradient,
envmap pre-training, TV / L1 / ortho regularisers, coarse-to-fine upsampling, and the multi-tensor Adam."""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from egonerf_amd.losses import TVLoss, ray_entropy_loss
from egonerf_amd.optim import FusedAdam
from tests.helpers import make_model, make_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture()
def extras(golden):
    fx = golden("train_extras")
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=int(fx["envmap_res_H"]))
    model = make_model(cfg, synth.make_weights(cfg, seed=int(fx["seed_weights"])), DEV)
    model.train()
    return fx, cfg, model


def _grad_of(model, k):
    if k == "envmap.emission":
        return model.envmap.emission.grad
    return dict(model.named_parameters())[k].grad


def _check(model, fx, prefix, tol):
    keys = [k[len(prefix) + 1:] for k in fx.files if k.startswith(prefix + "/")]
    assert len(keys) == 33
    bad = {}
    for k in keys:
        ref = fx[f"{prefix}/{k}"]
        g = _grad_of(model, k)
        g = np.zeros_like(ref) if g is None else g.detach().cpu().numpy()
        assert g.shape == ref.shape, k
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(g - ref).max()) / scale
        if err > tol and float(np.abs(ref).max()) > 0:
            bad[k] = err
        if float(np.abs(ref).max()) == 0:
            assert float(np.abs(g).max()) == 0, k
    assert not bad, bad


def _render(model, fx):
    return model(T(fx["rays"]), is_train=True, n_coarse=16, n_fine=16, exp_sampling=True, resampling=True, use_coarse_sample=True,
                 jitter=T(fx["jitter"]), u=T(fx["u"]))


def test_entropy_and_envmap_gradients(extras):
    fx, cfg, model = extras
    rgb, depth, bg, env, alpha = _render(model, fx)
    assert rgb.requires_grad and alpha.requires_grad and not depth.requires_grad  # depth: no_grad in the reference
    assert alpha.shape == (64, 33) and bool((alpha[:, -1] == 1).all())
    assert float((rgb.detach().cpu() - torch.from_numpy(fx["ent_rgb"])).abs().max()) <= 1e-4
    assert float((alpha.detach().cpu() - torch.from_numpy(fx["ent_alpha"])).abs().max()) <= 1e-4
    mse = torch.mean((rgb - T(fx["gt"])) ** 2)
    ent = ray_entropy_loss(alpha)
    assert abs(mse.item() - float(fx["ent_mse"])) <= 1e-6 and abs(ent.item() - float(fx["ent_entropy"])) <= 2e-5
    (mse + float(fx["entropy_weight"]) * ent).backward()
    _check(model, fx, "ent_grad", 2e-4)
    assert float(model.envmap.emission.grad.abs().max()) > 0


def test_entropy_alone_reaches_only_the_density_tables(extras):
    fx, cfg, model = extras
    alpha = _render(model, fx)[4]
    ray_entropy_loss(alpha).backward()
    _check(model, fx, "entonly_grad", 2e-4)
    for k, p in model.named_parameters():
        if not k.startswith("density_"):
            assert p.grad is None or float(p.grad.abs().max()) == 0, k


def test_ray_entropy_kernel_vs_torch_formula():
    g = torch.Generator().manual_seed(3)
    for N, S in ((1, 1), (5, 33), (130, 257), (64, 513)):
        a = torch.rand(N, S, generator=g).pow(4)
        a[0, : S // 2] = 0
        a_ref = a.double().requires_grad_(True)
        p = a_ref / (a_ref.sum(-1, keepdim=True) + 1e-10)
        ref = (-(p * torch.log2(p + 1e-10)).sum(-1)).mean()
        ref.backward()
        a_dev = a.to(DEV).requires_grad_(True)
        val = ray_entropy_loss(a_dev)
        (val * 3.0).backward()
        assert abs(val.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
        err = float((a_dev.grad.cpu().double() / 3.0 - a_ref.grad).abs().max())
        assert err <= 2e-5 * max(float(a_ref.grad.abs().max()), 1e-6), (N, S, err)


def test_envmap_pretraining_gradient(extras):
    fx, cfg, model = extras
    env = model(T(fx["rays"]), pretrain_envmap=True)
    assert env.requires_grad
    assert float((env.detach().cpu() - torch.from_numpy(fx["pre_env"])).abs().max()) <= 2e-6
    loss = torch.mean((env - T(fx["gt"])) ** 2)
    assert abs(loss.item() - float(fx["pre_loss"])) <= 1e-6
    loss.backward()
    g, ref = model.envmap.emission.grad.cpu().numpy(), fx["pre_grad"]
    assert float(np.abs(g - ref).max()) <= 1e-5 * float(np.abs(ref).max())
    with torch.no_grad():
        assert not model(T(fx["rays"]), pretrain_envmap=True).requires_grad


@pytest.mark.parametrize("name", ["tv_density", "tv_app", "l1", "ortho"])
def test_regularisers_value_and_gradient(extras, name):
    fx, cfg, model = extras
    tv = TVLoss()
    v = dict(tv_density=lambda: model.TV_loss_density(tv), tv_app=lambda: model.TV_loss_app(tv), l1=model.density_L1,
             ortho=model.vector_comp_diffs)[name]()
    ref = float(fx[f"reg/{name}/value"])
    assert v.dim() == 0 and abs(v.item() - ref) <= 5e-6 * max(abs(ref), 1.0)
    (v * 0.5).backward()
    keys = [k[len(f"reg/{name}/grad/"):] for k in fx.files if k.startswith(f"reg/{name}/grad/")]
    params = dict(model.named_parameters())
    for k in keys:
        g, r = params[k].grad.cpu().numpy() * 2.0, fx[f"reg/{name}/grad/{k}"]
        assert g.shape == r.shape
        assert float(np.abs(g - r).max()) <= 2e-5 * max(float(np.abs(r).max()), 1e-8), (name, k)
    assert all(p.grad is None for k, p in params.items() if k not in keys)


def test_tvloss_module_on_one_plane_and_full_size():
    """TVLoss()(plane) alone (utils.py:155-171), incl. a barbershop-size appearance plane against the torch formula."""
    g = torch.Generator().manual_seed(0)
    for C_, H, W in ((16, 10, 10), (48, 516, 150)):
        x = torch.randn(1, C_, H, W, generator=g)
        xr = x.double().requires_grad_(True)
        ref = 2 * ((xr[:, :, 1:] - xr[:, :, :-1]).pow(2).sum() / (C_ * (H - 1) * W) +
                   (xr[:, :, :, 1:] - xr[:, :, :, :-1]).pow(2).sum() / (C_ * H * (W - 1)))
        ref.backward()
        p = torch.nn.Parameter(x.to(DEV).contiguous(memory_format=torch.channels_last))
        v = TVLoss(TVLoss_weight=1)(p)
        v.backward()
        assert abs(v.item() - ref.item()) <= 2e-6 * ref.item()
        assert float((p.grad.cpu().double() - xr.grad).abs().max()) <= 1e-5 * float(xr.grad.abs().max())


def test_upsample_volume_grid(extras):
    fx, cfg, model = extras
    target = fx["up_target"].tolist()
    model.eval()
    before = model(T(fx["rays"]), n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)[0].clone()
    model.upsample_volume_grid(list(target))
    model.coordinates.set_resolution(list(target))  # train.py:376-377 (resets r0 to 0.05, coordinates.py:214)
    model.update_coarse_sigma_grid()
    assert model.coordinates.r0 == 0.05 and model.gridSize.tolist() == target
    sd = model.state_dict()
    for k in [k[3:] for k in fx.files if k.startswith("up/")]:
        assert tuple(sd[k].shape) == fx["up/" + k].shape, k
        assert sd[k].permute(0, 2, 3, 1).is_contiguous()
        assert float((sd[k].cpu() - torch.from_numpy(fx["up/" + k])).abs().max()) <= 5e-6, k
    rgb, depth, *_ = model(T(fx["rays"]), n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)
    assert float((rgb.detach().cpu() - torch.from_numpy(fx["up_rgb"])).abs().max()) <= 1e-4
    assert float((depth.detach().cpu() - torch.from_numpy(fx["up_depth"])).abs().max()) <= 1e-3
    assert float((rgb.detach() - before.detach()).abs().max()) > 1e-3  # the scene cache noticed the new tables / LUT / schedule


def test_fused_adam_matches_torch_adam():
    """Several steps with per-group lr and the reference's per-step lr decay (train.py:328-329) vs torch.optim.Adam on CPU."""
    g = torch.Generator().manual_seed(1)
    shapes = [(1, 16, 10, 12), (1, 48, 30, 1), (27, 144), (128,), (3, 128), (1025,), (3, 32, 16)]
    cpu = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    dev = []
    for t in cpu:
        d = t.detach().to(DEV)
        if d.dim() == 4:
            d = d.contiguous(memory_format=torch.channels_last)
        dev.append(d.requires_grad_(True))
    groups = lambda ps: [dict(params=ps[:2], lr=0.02), dict(params=ps[2:5], lr=1e-3), dict(params=ps[5:], lr=0.1)]
    o_ref, o_dev = torch.optim.Adam(groups(cpu), betas=(0.9, 0.99)), FusedAdam(groups(dev), betas=(0.9, 0.99))
    for it in range(6):
        for a, b in zip(cpu, dev):
            gr = torch.randn(a.shape, generator=g) * (10.0 ** -(it % 4))
            if it == 3:
                gr[..., ::2] = 0
            a.grad = gr
            b.grad = gr.to(DEV)  # standard-contiguous gradient for a channel-last parameter: the optimiser re-strides it
        o_ref.step(), o_dev.step()
        for grp_r, grp_d in zip(o_ref.param_groups, o_dev.param_groups):
            grp_r["lr"] *= 0.977
            grp_d["lr"] *= 0.977
        for a, b in zip(cpu, dev):
            assert float((a.detach() - b.detach().cpu()).abs().max()) <= 2e-6, (it, a.shape)
    assert dev[0]._version >= 6


def test_training_loop_body_reads_like_the_reference(extras):
    """train.py:245-330 assembled: render + MSE + TV + L1 + ortho + entropy, backward, FusedAdam, lr decay — the loss falls."""
    fx, cfg, model = extras
    opt = FusedAdam(model.get_optparam_groups(0.02, 1e-3, 0.005), betas=(0.9, 0.99))
    tv, lr_factor, gt = TVLoss(), 0.1 ** (1 / 100), T(fx["gt"])
    losses = []
    for it in range(12):
        rgb, depth, _, _, alpha = _render(model, fx)
        loss = torch.mean((rgb - gt) ** 2)
        total = loss + 1e-4 * model.vector_comp_diffs() + 1e-4 * model.density_L1() + 0.1 * model.TV_loss_density(tv) \
            + 0.01 * model.TV_loss_app(tv) + 1e-3 * ray_entropy_loss(alpha)
        opt.zero_grad()
        total.backward()
        opt.step()
        for grp in opt.param_groups:
            grp["lr"] *= lr_factor
        model.update_coarse_sigma_grid()
        losses.append(loss.item())
    assert losses[-1] < 0.6 * losses[0], losses
