"""GPU: the training iteration as a replayed hipGraph (egonerf_amd.train.GraphedTrainStep) against the eager loop body of
train.py:245-330, and the device-clock Adam (`FusedAdam(capturable=True)`, ego_adam_step_graph) against torch.optim.Adam with
the reference's per-step learning-rate decay."""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from egonerf_amd.optim import FusedAdam
from egonerf_amd.train import GraphedTrainStep
from tests.helpers import make_model

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(n_coarse=16, n_fine=16, exp_sampling=True, resampling=True, use_coarse_sample=True)


def test_device_clock_adam_matches_torch_adam_with_lr_decay():
    g = torch.Generator().manual_seed(5)
    shapes = [(1, 16, 10, 12), (27, 144), (128,), (1025,)]
    cpu = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    dev = [t.detach().to(DEV).requires_grad_(True) for t in cpu]
    groups = lambda ps: [dict(params=ps[:1], lr=0.02), dict(params=ps[1:], lr=1e-3)]
    factor = 0.977
    o_ref = torch.optim.Adam(groups(cpu), betas=(0.9, 0.99))
    o_dev = FusedAdam(groups(dev), betas=(0.9, 0.99), capturable=True, lr_factor=factor)
    for it in range(8):
        for a, b in zip(cpu, dev):
            a.grad = torch.randn(a.shape, generator=g) * (10.0 ** -(it % 3))
            b.grad = a.grad.to(DEV)
        o_ref.step(), o_dev.step()
        for grp in o_ref.param_groups:
            grp["lr"] *= factor  # train.py:328-329; the capturable optimiser does this on the device
        for a, b in zip(cpu, dev):
            assert float((a.detach() - b.detach().cpu()).abs().max()) <= 2e-6, (it, a.shape)
    assert abs(o_dev.lr_scale() - factor ** 8) <= 1e-12
    assert [grp["lr"] for grp in o_dev.param_groups] == [0.02, 1e-3]  # base rates stay


def _same_after_adam(name, a, b, n_steps, lr):
    """Two runs of the same iterations differ by the order of the table scatters' float atomics: gradients agree to rounding,
    but Adam's m / (sqrt(v) + eps) turns the rounding of a gradient element whose contributions nearly cancel into an O(lr)
    difference of that one element (seen in two of sixteen sessions: 5e-4 on one appearance texel).  So: the tensors agree closely
    on average and in all but a handful of elements, and no element differs by more than the steps could have moved it."""
    scale = max(float(a.abs().max()), 1e-3)
    err = (a - b).abs()
    # tools/graph_vs_eager_stress.py, 40 repetitions: mean <= 1.1e-6 of the scale; in 4 of them the same 1 + 4 texels of two
    # 14 400-element appearance planes are 5e-4 / 7e-4 off; a wrong step size, a stale buffer or a skipped iteration would put
    # EVERY element O(lr) = 1e-2 off
    assert float(err.mean()) <= 1e-5 * scale, (name, float(err.mean()))
    n_off = int((err > 2e-4 * scale).sum())
    assert n_off <= 8 + 1e-3 * err.numel(), (name, n_off, float(err.max()))
    assert float(err.max()) <= 2.0 * n_steps * lr, (name, float(err.max()))


def _setup(seed, **cfg_kw):
    cfg = synth.SceneConfig(n_voxel=20 ** 3, **cfg_kw)
    model = make_model(cfg, synth.make_weights(cfg, seed=seed), DEV)
    model.train()
    model.update_coarse_sigma_grid()
    return cfg, model


def test_graphed_step_equals_eager_loop_body():
    """Six iterations on changing ray batches with pinned is_train noise: replayed graph vs the eager loop (same kernels; the table
    scatters use float atomics, so parameters agree to rounding, not bit for bit)."""
    N, factor = 192, 0.9
    batches = [(torch.from_numpy(synth.make_rays(N, seed=40 + i)).to(DEV),
                torch.from_numpy(synth.hash_uniform(70 + i, 0, N * 3).reshape(N, 3).astype(np.float32)).to(DEV)) for i in range(6)]
    jit = torch.from_numpy(synth.hash_uniform(9, 0, N * 16).reshape(N, 16).astype(np.float32)).to(DEV)
    noise = lambda n, m, dev: jit
    # eager reference
    _, m_ref = _setup(3)
    o_ref = FusedAdam(m_ref.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    ref_losses = []
    for rays, gt in batches:
        rgb, *_ = m_ref(rays, is_train=True, jitter=jit, u=jit, **KW)
        loss = torch.mean((rgb - gt) ** 2)
        o_ref.zero_grad(set_to_none=True)
        loss.backward()
        o_ref.step()
        for grp in o_ref.param_groups:
            grp["lr"] *= factor
        m_ref.update_coarse_sigma_grid()
        ref_losses.append(float(loss.detach()))
    # graphed: the constructor trains on batch 0 (warmup = 1), the replays take batches 1..5
    _, m_g = _setup(3)
    o_g = FusedAdam(m_g.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), capturable=True, lr_factor=factor)
    step = GraphedTrainStep(m_g, o_g, batches[0][0], batches[0][1], KW, warmup=1, noise_fn=noise)
    got = [float(step(rays, gt)) for rays, gt in batches[1:]]
    assert step.iterations == 6
    for a, b in zip(ref_losses[1:], got):
        assert abs(a - b) <= 2e-5 * max(abs(a), 1e-3), (ref_losses, got)
    pr, pg = dict(m_ref.named_parameters()), dict(m_g.named_parameters())
    for k in pr:
        _same_after_adam(k, pr[k].detach(), pg[k].detach(), n_steps=6, lr=0.02)
    # an eager render after the replays sees the CURRENT weights (the packed-weight cache follows the version bump)
    m_ref.eval(); m_g.eval()
    with torch.no_grad():
        a = m_ref(batches[0][0], n_coarse=32, exp_sampling=True)[0]
        b = m_g(batches[0][0], n_coarse=32, exp_sampling=True)[0]
    # (a stale packed blob would be one optimiser step behind: ~1e-2 in the image; an element-level Adam outlier as described in
    # _same_after_adam moves a pixel by far less)
    assert float((a - b).abs().max()) <= 1e-3


def test_graphed_step_with_envmap_and_regularisers():
    """The Ricoh-style iteration (configs/EgoNeRF/ricoh/common.txt:12-13 + opt.py defaults): envmap parameter, TV / L1 / ortho /
    ray-entropy terms inside the captured loss - every one of those nodes is asynchronous, so the capture must take them."""
    from egonerf_amd.losses import TVLoss, ray_entropy_loss
    N, factor = 128, 0.95
    tv = TVLoss()
    batches = [(torch.from_numpy(synth.make_rays(N, seed=50 + i)).to(DEV),
                torch.from_numpy(synth.hash_uniform(90 + i, 0, N * 3).reshape(N, 3).astype(np.float32)).to(DEV)) for i in range(4)]
    jit = torch.from_numpy(synth.hash_uniform(19, 0, N * 16).reshape(N, 16).astype(np.float32)).to(DEV)

    def make():
        _, m = _setup(6, use_envmap=True, envmap_res_H=16)
        loss_fn = lambda rgb, gt, alpha: (torch.mean((rgb - gt) ** 2) + 1e-4 * m.vector_comp_diffs() + 8e-5 * m.density_L1()
                                          + 0.1 * m.TV_loss_density(tv) + 0.01 * m.TV_loss_app(tv) + 1e-3 * ray_entropy_loss(alpha))
        return m, loss_fn

    m_ref, loss_ref = make()
    o_ref = FusedAdam(m_ref.get_optparam_groups(0.02, 1e-3, 0.01), betas=(0.9, 0.99))
    ref_losses = []
    for rays, gt in batches:
        rgb, _d, _bg, _env, alpha = m_ref(rays, is_train=True, jitter=jit, u=jit, **KW)
        loss = loss_ref(rgb, gt, alpha)
        o_ref.zero_grad(set_to_none=True)
        loss.backward()
        o_ref.step()
        for grp in o_ref.param_groups:
            grp["lr"] *= factor
        m_ref.update_coarse_sigma_grid()
        ref_losses.append(float(loss.detach()))
    m_g, loss_g = make()
    o_g = FusedAdam(m_g.get_optparam_groups(0.02, 1e-3, 0.01), betas=(0.9, 0.99), capturable=True, lr_factor=factor)
    step = GraphedTrainStep(m_g, o_g, batches[0][0], batches[0][1], KW, loss_fn=loss_g, warmup=1, noise_fn=lambda n, m, dev: jit)
    got = [float(step(rays, gt)) for rays, gt in batches[1:]]
    for a, b in zip(ref_losses[1:], got):
        assert abs(a - b) <= 5e-5 * max(abs(a), 1e-3), (ref_losses, got)
    _same_after_adam("envmap.emission", m_ref.envmap.emission.detach(), m_g.envmap.emission.detach(), n_steps=4, lr=0.01)
    assert float((m_ref.envmap.emission.detach() - make()[0].envmap.emission.detach()).abs().max()) > 0  # the envmap did train


def test_graphed_step_needs_the_capturable_optimiser():
    _, model = _setup(4)
    opt = FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    rays = torch.from_numpy(synth.make_rays(64, seed=1)).to(DEV)
    with pytest.raises(ValueError, match="capturable"):
        GraphedTrainStep(model, opt, rays, torch.zeros(64, 3, device=DEV), KW)


def test_graphed_step_with_decaying_regulariser_weights():
    """train.py:295-309 multiplies the TV weights (while iteration < iter_ignore_TV) and the ray-entropy weight
    (once iteration > iter_ignore_entropy) by lr_factor in every active iteration.  A Python float would be frozen into the graph at
    capture; the graphed loop takes them from the device-side schedule (GraphedTrainStep.schedule.decayed).  Seven iterations with
    both gates switching inside the run, eager loop with the reference's own Python-side bookkeeping vs replays."""
    from egonerf_amd.losses import TVLoss, ray_entropy_loss
    N, factor, n_it = 128, 0.8, 7          # a strong decay so that a frozen weight would be far off
    ignore_tv, ignore_entropy = 4, 2        # TV active in iterations 0..3, entropy in iterations 3..6
    tv = TVLoss()
    batches = [(torch.from_numpy(synth.make_rays(N, seed=60 + i)).to(DEV),
                torch.from_numpy(synth.hash_uniform(110 + i, 0, N * 3).reshape(N, 3).astype(np.float32)).to(DEV)) for i in range(n_it)]
    jit = torch.from_numpy(synth.hash_uniform(29, 0, N * 16).reshape(N, 16).astype(np.float32)).to(DEV)
    W_TVD, W_TVA, W_ENT = 1.0, 0.1, 0.05
    # eager loop, the reference's bookkeeping verbatim (train.py:295-309)
    _, m_ref = _setup(8)
    o_ref = FusedAdam(m_ref.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    w_tvd, w_tva, w_ent = W_TVD, W_TVA, W_ENT
    ref_losses, ref_weights = [], []
    for it, (rays, gt) in enumerate(batches):
        rgb, _d, _bg, _env, alpha = m_ref(rays, is_train=True, jitter=jit, u=jit, **KW)
        loss = torch.mean((rgb - gt) ** 2)
        if it < ignore_tv:
            w_tvd *= factor
            loss = loss + m_ref.TV_loss_density(tv) * w_tvd
            w_tva *= factor
            loss = loss + m_ref.TV_loss_app(tv) * w_tva
        if it > ignore_entropy:
            w_ent *= factor
            loss = loss + ray_entropy_loss(alpha) * w_ent
        ref_weights.append((w_tvd if it < ignore_tv else 0.0, w_ent if it > ignore_entropy else 0.0))
        o_ref.zero_grad(set_to_none=True)
        loss.backward()
        o_ref.step()
        for grp in o_ref.param_groups:
            grp["lr"] *= factor
        m_ref.update_coarse_sigma_grid()
        ref_losses.append(float(loss.detach()))
    # graphed loop: weights from the device-side schedule
    _, m_g = _setup(8)
    o_g = FusedAdam(m_g.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), capturable=True, lr_factor=factor)
    seen = []

    def loss_fn(rgb, gt, alpha, sched):
        wd = sched.decayed(W_TVD, factor, active_before=ignore_tv)
        wa = sched.decayed(W_TVA, factor, active_before=ignore_tv)
        we = sched.decayed(W_ENT, factor, active_after=ignore_entropy)
        seen.append((wd, we))
        return torch.mean((rgb - gt) ** 2) + m_g.TV_loss_density(tv) * wd + m_g.TV_loss_app(tv) * wa + ray_entropy_loss(alpha) * we

    step = GraphedTrainStep(m_g, o_g, batches[0][0], batches[0][1], KW, loss_fn=loss_fn, warmup=1, noise_fn=lambda n, m, dev: jit)
    got = []
    for it, (rays, gt) in enumerate(batches[1:], start=1):
        got.append(float(step(rays, gt)))
        wd, we = (float(t) for t in seen[-1])    # the captured weight tensors hold this replay's values
        assert abs(wd - ref_weights[it][0]) <= 1e-6 and abs(we - ref_weights[it][1]) <= 1e-7, (it, wd, we, ref_weights[it])
    assert step.schedule.iteration() == n_it
    for a, b in zip(ref_losses[1:], got):
        assert abs(a - b) <= 5e-5 * max(abs(a), 1e-3), (ref_losses, got)
    pr, pg = dict(m_ref.named_parameters()), dict(m_g.named_parameters())
    for k in pr:
        _same_after_adam(k, pr[k].detach(), pg[k].detach(), n_steps=n_it, lr=0.02)


def test_capturable_adam_state_dict_carries_the_device_clock():
    """the step count and lr scale of capturable mode live in a device buffer; state_dict() / load_state_dict()
    must carry them, or a resumed run restarts bias correction and decay at t = 0."""
    g = torch.Generator().manual_seed(7)
    mk = lambda: [torch.randn(33, 5, generator=g).to(DEV).requires_grad_(True), torch.randn(64, generator=g).to(DEV).requires_grad_(True)]
    g.manual_seed(7); a = mk()
    g.manual_seed(7); b = mk()
    grads = [[torch.randn(p.shape, generator=g).to(DEV) for p in a] for _ in range(6)]
    oa = FusedAdam([dict(params=a, lr=0.01)], betas=(0.9, 0.99), capturable=True, lr_factor=0.9)
    ob = FusedAdam([dict(params=b, lr=0.01)], betas=(0.9, 0.99), capturable=True, lr_factor=0.9)
    for k in range(3):
        for p, q, gr in zip(a, b, grads[k]):
            p.grad, q.grad = gr, gr.clone()
        oa.step(), ob.step()
    sd = ob.state_dict()
    assert sd["ego_clock"][0] == 3.0 and abs(sd["ego_clock"][1] - 0.9 ** 3) < 1e-12
    assert all(st["step"] == 3 for st in sd["state"].values())
    oc = FusedAdam([dict(params=b, lr=0.01)], betas=(0.9, 0.99), capturable=True, lr_factor=0.5)   # a fresh optimiser on the same tensors
    oc.load_state_dict(sd)
    assert oc.steps_taken() == 3 and abs(oc.lr_scale() - 0.9 ** 3) < 1e-12 and oc.lr_factor == 0.9
    assert abs(oc.current_lrs()[0] - 0.01 * 0.9 ** 3) < 1e-12
    for k in range(3, 6):
        for p, q, gr in zip(a, b, grads[k]):
            p.grad, q.grad = gr, gr.clone()
        oa.step(), oc.step()
    for p, q in zip(a, b):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-7
    with pytest.warns(RuntimeWarning, match="BASE rate"):
        oc.param_groups[0]["lr"] *= 0.5
        for q, gr in zip(b, grads[0]):
            q.grad = gr
        oc.step()


def test_capturable_adam_loads_a_checkpoint_written_without_the_device_clock():
    """a state dict saved in non-capturable mode (the caller decays group['lr'] itself, train.py:328-329) loaded into
    a capturable optimiser must continue at the saved step count with the saved (already decayed) rates as base rates - not restart
    the bias correction at t = 0."""
    g = torch.Generator().manual_seed(11)
    mk = lambda: [torch.randn(17, 9, generator=g).to(DEV).requires_grad_(True), torch.randn(40, generator=g).to(DEV).requires_grad_(True)]
    g.manual_seed(11); a = mk()
    g.manual_seed(11); b = mk()
    grads = [[torch.randn(p.shape, generator=g).to(DEV) for p in a] for _ in range(6)]
    factor = 0.9
    oa = FusedAdam([dict(params=a, lr=0.01)], betas=(0.9, 0.99))     # eager reference: six steps, host-side decay
    ob = FusedAdam([dict(params=b, lr=0.01)], betas=(0.9, 0.99))
    for k in range(3):
        for p, q, gr in zip(a, b, grads[k]):
            p.grad, q.grad = gr, gr.clone()
        oa.step(), ob.step()
        for o in (oa, ob):
            o.param_groups[0]["lr"] *= factor
    sd = ob.state_dict()
    assert "ego_clock" not in sd
    oc = FusedAdam([dict(params=b, lr=0.01)], betas=(0.9, 0.99), capturable=True, lr_factor=factor)
    oc.load_state_dict(sd)
    assert oc.steps_taken() == 3 and oc.lr_scale() == 1.0 and abs(oc.current_lrs()[0] - 0.01 * factor ** 3) < 1e-12
    for k in range(3, 6):
        for p, q, gr in zip(a, b, grads[k]):
            p.grad, q.grad = gr, gr.clone()
        oa.step(), oc.step()
        oa.param_groups[0]["lr"] *= factor
    for p, q in zip(a, b):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-7


def test_graphed_step_passes_the_schedule_only_on_opt_in():
    """no arity sniffing - a fourth parameter with an unrelated default keeps its default, a *args loss gets the
    schedule with schedule_aware=True, a parameter named `sched` opts in by itself."""
    from egonerf_amd.train import TrainSchedule
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    rays = torch.from_numpy(synth.make_rays(64, seed=2)).to(DEV)
    gt = torch.rand(64, 3, device=DEV)
    kw = dict(n_coarse=16, exp_sampling=True)
    seen = {}

    def make(loss_fn, **extra):
        m = make_model(cfg, synth.make_weights(cfg, seed=9), DEV)
        m.train()
        o = FusedAdam(m.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), capturable=True)
        return GraphedTrainStep(m, o, rays, gt, kw, loss_fn=loss_fn, warmup=1, **extra)

    def with_default(rgb, tgt, alpha, reduction="mean"):
        seen["reduction"] = reduction
        return torch.mean((rgb - tgt) ** 2)

    def star(*args):
        seen["n_args"] = len(args)
        seen["last"] = args[-1]
        return torch.mean((args[0] - args[1]) ** 2)

    def named(rgb, tgt, alpha, sched):
        seen["named"] = sched
        return torch.mean((rgb - tgt) ** 2)

    make(with_default)
    assert seen["reduction"] == "mean"
    make(star)
    assert seen["n_args"] == 3
    make(star, schedule_aware=True)
    assert seen["n_args"] == 4 and isinstance(seen["last"], TrainSchedule)
    make(named)
    assert isinstance(seen["named"], TrainSchedule)
