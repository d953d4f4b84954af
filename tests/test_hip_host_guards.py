"""GPU: host-layer guards found by review - per-parameter re-allocation (`Module._apply`) vs the kernels' compact
`base + 32-bit offset` addressing, and the f16f8 arithmetic's behaviour on large / tiny activations."""
import ctypes

import numpy as np
import pytest
import torch

from egonerf_amd import _lib, synth
from tests.helpers import make_model, make_oracle, maxerr

pytestmark = pytest.mark.gpu


def _render(model, rays):
    with torch.no_grad():
        return model(rays, n_coarse=24, n_fine=24, resampling=True, exp_sampling=True)


def test_model_built_on_cpu_then_moved_renders_and_stays_compact():
    """model.to('cuda') / .cuda() / .float() re-allocate every parameter separately (nn.Module._apply); the model re-carves each
    field's 12 tables from one buffer, keeps the Parameter objects (an optimiser's references stay valid) and rebuilds its
    non-parameter device state (pooled density tables)."""
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    w = synth.make_weights(cfg, seed=1234)
    ref_model = make_model(cfg, w, "cuda")
    rays = torch.from_numpy(synth.make_rays(200, seed=3)).cuda()
    want = _render(ref_model, rays)
    cpu_model = make_model(cfg, w, "cpu")
    with pytest.raises(RuntimeError):
        cpu_model.scene()                       # no CPU fallback
    ids = [id(p) for p in cpu_model.parameters()]
    moved = cpu_model.to("cuda")
    assert [id(p) for p in moved.parameters()] == ids
    assert moved._is_compact("density") and moved._is_compact("app")
    got = _render(moved, rays)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    again = _render(moved.float().cuda(), rays)   # no-op conversions must not break anything either
    assert torch.equal(again[0], want[0])


def test_non_compact_tables_are_recompacted_lazily():
    """Parameters assigned one by one (what load_state_dict(assign=True) does): scene() notices a field that is not within one
    4 GB window and re-carves it; here the window is shrunk artificially by planting one table far away is not possible in a
    test, so the hook is exercised through _recompact_tables directly and the render compared."""
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    w = synth.make_weights(cfg, seed=77)
    model = make_model(cfg, w, "cuda")
    rays = torch.from_numpy(synth.make_rays(64, seed=5)).cuda()
    want = _render(model, rays)
    ids = [id(p) for p in model.parameters()]
    for l in model._table_lists("app"):      # separate allocations, as after assign=True
        for p in l:
            p.data = p.data.clone(memory_format=torch.preserve_format)
    model._recompact_tables("app")
    model._recompact_tables("density")
    assert [id(p) for p in model.parameters()] == ids and model._is_compact("app")
    got = _render(model, rays)
    assert torch.equal(got[0], want[0])


@pytest.mark.parametrize("gain", [1.0, 6.0, 40.0])
def test_f16f8_f16f6_on_large_and_tiny_activations(gain):
    """The f16f8 arithmetic keeps an e4m3 copy of every activation and of w_hi for the two low-order product terms:
    e4m3 saturates at 448 (MODE.FP16_OVFL, set by the kernel) and flushes below 2^-9.  Scale the MLP weights so that hidden
    activations reach hundreds to thousands (gain 6 / 40) and check the per-sample colour against the fp32-MFMA kernel: a
    saturated correction operand may cost accuracy of a low-order term only, never a NaN or an O(1) error.  The default f16f6
    arithmetic scales every block of 32 values by its own largest magnitude, so it has no fixed range; it is held to the same bounds."""
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    w = synth.make_weights(cfg, seed=1234, mlp_gain=3.0 * gain)
    model = make_model(cfg, w, "cuda")
    M = 4096
    feat = torch.from_numpy((synth.hash_normal(9, 1, M * 27).reshape(M, 27) * (2.0 if gain > 1 else 1.0)).astype(np.float32)).cuda()
    feat[: M // 8] *= 1e-4                       # tiny activations: the fp8 copies flush to zero, the fp16 main term stays
    dirs = torch.from_numpy(synth.make_rays(M, seed=2)[:, 3:6].copy()).cuda()
    out = {}
    for prec in ("f32", "f16x3", "f16f8", "f16f6"):
        model.mlp_precision = prec
        out[prec] = model.renderModule(None, dirs, feat)
        assert bool(torch.isfinite(out[prec]).all())
    e3, e8, e6 = maxerr(out["f16x3"], out["f32"]), maxerr(out["f16f8"], out["f32"]), maxerr(out["f16f6"], out["f32"])
    print(f"gain {gain}: max |d rgb| per sample vs the fp32-MFMA kernel: f16x3 {e3:.2e}, f16f8 {e8:.2e}, f16f6 {e6:.2e}")
    # Errors are relative to the magnitudes inside the MLP (2^-21 / ~2^-16 per product), so per-sample colour errors grow with the
    # weight gain until the sigmoid saturates; what must never happen is a NaN / inf or an O(1) error from a saturated or flushed
    # fp8 operand.  Bounds = measured values with ~2.5x margin: gain 1 (the bench scene's weights): 9.5e-7 / 2.6e-5; gain 6 (hidden
    # activations in the hundreds, logits of +-50): 1.2e-4 / 3.3e-3 - the f16f8 arithmetic is 2^-16-relative, so a checkpoint with such
    # logits must be rendered with mlp_precision = 'f16x3' (EgoNeRF.check_mlp_precision measures it on sample rays); gain 40: the
    # sigmoid saturates, 2.4e-8 / 1.7e-6.  Composited errors are ~3x smaller than per-sample ones
    assert e3 <= {1.0: 3e-6, 6.0: 3e-4, 40.0: 1e-6}[gain], (gain, e3)
    assert e8 <= {1.0: 6e-5, 6.0: 8e-3, 40.0: 1e-4}[gain], (gain, e8)
    assert e6 <= {1.0: 6e-5, 6.0: 8e-3, 40.0: 1e-4}[gain], (gain, e6)


def test_activations_beyond_the_half_range_do_not_poison_a_pixel():
    """hidden activations above 65504 used to turn into inf in the fp16 split (inf - inf = NaN in the residual) and in
    the training forward's half-precision dumps.  With MODE.FP16_OVFL set for every arithmetic the main term saturates and the residual
    carries the rest (exact up to twice the half range): colours stay finite and equal the oracle's (saturated sigmoids) in all four
    arithmetics, and a training step's gradients are finite."""
    import numpy as np
    from egonerf_amd import synth
    from tests.helpers import make_model, make_oracle
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    w = synth.make_weights(cfg, seed=7)
    w["renderModule.mlp.0.weight"] = (w["renderModule.mlp.0.weight"] * 4000.0).astype(np.float32)   # layer-1 activations of ~1e5
    model, oracle = make_model(cfg, w, "cuda"), make_oracle(cfg, w)
    rays = torch.from_numpy(synth.make_rays(64, seed=3))
    ref = oracle.forward(rays, n_coarse=32)
    for prec in ("f16x3", "f16f8", "f16f6", "f32"):
        model.mlp_precision = prec
        with torch.no_grad():
            got = model(rays.cuda(), n_coarse=32, exp_sampling=True)
        assert bool(torch.isfinite(got[0]).all()), prec
        # the hidden layer is far outside any fp16-grade accuracy claim here: the point is "no NaN" everywhere, the f32 arithmetic exact, and
        # the three-term split still sane (f16f8's fixed scales saturate at 448 by design: model.check_mlp_precision is the guard for that)
        if prec in ("f32", "f16x3"):
            assert float((got[0].cpu() - ref[0]).abs().max()) <= (1e-4 if prec == "f32" else 0.35), prec
    model.train()
    rgb, *_ = model(rays.cuda(), is_train=True, n_coarse=32, exp_sampling=True)
    torch.mean(rgb ** 2).backward()
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters())


def test_copy_out_moves_any_size_and_alignment_and_refuses_bad_arguments():
    """ABI v17 ego_copy_out: up to EGO_COPY_OUT_MAX float arrays device -> mapped (pinned) host memory in one small launch: aligned and
    misaligned pointers, sizes that are no multiple of four floats, empty arrays, one workgroup and many; bad arguments are refused."""
    import ctypes as C
    from egonerf_amd import _lib
    lib, st = _lib.load(), _lib.stream_handle()
    g = torch.Generator().manual_seed(3)
    sizes = [0, 1, 3, 4, 5, 1023, 4096 * 3, 70001]
    for wgs in (1, 4, 64):
        big_d = torch.rand(sum(sizes) + 64, generator=g).cuda()
        big_h = torch.full((sum(sizes) + 64,), -1.0).pin_memory()
        src, dst, off = [], [], 0
        for k, n in enumerate(sizes):
            o = off + (k % 3)   # every third pair 16-byte aligned relative to the base, the others not
            src.append(big_d[o:o + n]); dst.append(big_h[o:o + n]); off = o + n
        n = len(sizes)
        ps = (C.c_void_p * n)(*[t.data_ptr() if t.numel() else None for t in src])
        pd = (C.c_void_p * n)(*[t.data_ptr() if t.numel() else None for t in dst])
        cn = (C.c_int64 * n)(*sizes)
        _lib.check(lib.ego_copy_out(n, ps, pd, cn, wgs, st), "ego_copy_out")
        torch.cuda.synchronize()
        want = torch.full_like(big_h, -1.0)
        for s_, d_ in zip(src, dst):
            want[d_.storage_offset():d_.storage_offset() + d_.numel()] = s_.cpu()
        assert torch.equal(big_h, want), wgs   # every array arrived, nothing next to them was touched
    one = (C.c_void_p * 1)(big_d.data_ptr()); oneh = (C.c_void_p * 1)(big_h.data_ptr()); c1 = (C.c_int64 * 1)(8)
    assert lib.ego_copy_out(0, None, None, None, 4, st) == 0
    assert lib.ego_copy_out(1, one, oneh, c1, 0, st) != 0          # no workgroup
    assert lib.ego_copy_out(9, one, oneh, c1, 4, st) != 0          # more than EGO_COPY_OUT_MAX arrays
    assert lib.ego_copy_out(1, one, (C.c_void_p * 1)(None), c1, 4, st) != 0
    assert lib.ego_copy_out(1, one, oneh, (C.c_int64 * 1)(-1), 4, st) != 0


@pytest.mark.parametrize("copy_form", ["kernel_delayed", "kernel", "runtime"])
def test_host_hand_over_forms_agree(copy_form, monkeypatch):
    """The three forms of the device -> host copies (ego_copy_out under the NEXT chunk's shade kernel = the default, ego_copy_out as soon
    as the chunk is done, the runtime's hipMemcpyAsync) return the same bits; an odd chunk size puts the rows of the pinned arrays at
    addresses that are no multiple of 16."""
    from egonerf_amd import renderer
    monkeypatch.setattr(renderer, "_COPY_WORKGROUPS", 0 if copy_form == "runtime" else 4)
    monkeypatch.setattr(renderer, "_DELAY_COPIES", copy_form == "kernel_delayed")
    cfg = synth.SceneConfig(n_voxel=24 ** 3)
    model = make_model(cfg, synth.make_weights(cfg, seed=78), "cuda")
    host = torch.from_numpy(synth.make_rays(1000, seed=10))
    kw = dict(chunk=193, n_coarse=33, exp_sampling=True, device="cuda")
    with torch.no_grad():
        want = renderer.volume_renderer(host.cuda(), model, keep_alpha=True, **kw)
        got = renderer.volume_renderer(host, model, keep_alpha=True, empty_gpu_cache=True, **kw)
    for j, (w_, g_) in enumerate(zip(want, got)):
        assert (w_ is None) == (g_ is None), j
        if w_ is not None:
            assert np.array_equal(g_, w_.cpu().numpy()), (copy_form, j)


@pytest.mark.parametrize("where", ["pageable", "pinned", "device"])
def test_volume_renderer_host_hand_over_equals_the_resident_path(where):
    """the reference's own call pattern - volume_renderer(..., empty_gpu_cache=True), every chunk's outputs (the
    [chunk, S + 1] alpha included) copied back (renderer.py:26, :39-53) - through pinned staging and a copy stream that runs under the
    next chunk's kernels.  Same kernels per chunk, so the host arrays must equal the resident tensors bit for bit; ragged last chunk,
    an envmap scene (all five outputs present), keep_alpha on and off, and an empty ray list."""
    from egonerf_amd.renderer import volume_renderer
    cfg = synth.SceneConfig(n_voxel=24 ** 3, use_envmap=True, envmap_res_H=16)
    model = make_model(cfg, synth.make_weights(cfg, seed=77), "cuda")
    host = torch.from_numpy(synth.make_rays(1000, seed=9))
    rays = {"pageable": host, "pinned": host.pin_memory(), "device": host.cuda()}[where]
    kw = dict(chunk=192, n_coarse=32, n_fine=32, resampling=True, exp_sampling=True, device="cuda")
    with torch.no_grad():
        for keep_alpha in (True, False):
            want = volume_renderer(host.cuda(), model, keep_alpha=keep_alpha, **kw)
            got = volume_renderer(rays, model, keep_alpha=keep_alpha, empty_gpu_cache=True, **kw)
            assert len(got) == 5
            for j, (w_, g_) in enumerate(zip(want, got)):
                assert (w_ is None) == (g_ is None), j
                if w_ is not None:
                    assert isinstance(g_, np.ndarray) and g_.shape == tuple(w_.shape), (j, g_.shape, w_.shape)
                    assert np.array_equal(g_, w_.cpu().numpy()), j
            assert (got[4] is None) == (not keep_alpha)
        empty = volume_renderer(rays[:0], model, empty_gpu_cache=True, **kw)
        assert empty[0].shape == (0, 3) and empty[1].shape == (0,)


def test_marched_event_is_recorded_on_every_path():
    """ABI v17 ego_render_args.marched / EgoNeRF.forward(marched_event=ev): the event is recorded behind the call's last march (render path)
    or at the end of a call that has no march of its own (an empty batch, the envmap pre-training branch, a differentiable call) - a
    stream that waits for it must never wait for a record that does not come - and passing it changes no output bit."""
    cfg = synth.SceneConfig(n_voxel=24 ** 3, use_envmap=True, envmap_res_H=16)
    model = make_model(cfg, synth.make_weights(cfg, seed=79), "cuda")
    rays = torch.from_numpy(synth.make_rays(200, seed=11)).cuda()
    kw = dict(n_coarse=32, n_fine=32, resampling=True, exp_sampling=True)
    other = torch.cuda.Stream()
    with torch.no_grad():
        want = model(rays, **kw)
        for r, extra in ((rays, {}), (rays[:0], {}), (rays, dict(pretrain_envmap=True))):
            ev = torch.cuda.Event()
            ev.record()   # (creates the underlying event: the library records it by handle)
            torch.cuda.synchronize()
            got = model(r, marched_event=ev, **kw, **extra)
            other.wait_event(ev)
            other.synchronize()   # returns: the wait found a record to wait for
            assert ev.query()
            if r.shape[0] and not extra:
                for a_, b_ in zip(want, got):
                    assert (a_ is None) == (b_ is None) and (a_ is None or torch.equal(a_, b_))
    model.train()
    ev = torch.cuda.Event()
    ev.record()
    rgb = model(rays, is_train=True, marched_event=ev, **kw)[0]
    assert rgb.requires_grad
    other.wait_event(ev)
    other.synchronize()
    assert ev.query()
