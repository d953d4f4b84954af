"""CPU-only checks: the C-ABI library loads and exports every declared symbol (no compute calls), host
logic (LUTs, schedules, resolution rule, samplers, parameter layout, sharding) and the failure mode
without a GPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from egonerf_amd import _lib, synth
from egonerf_amd.coordinates import YinYangSphericalCoords
from egonerf_amd.renderer import psnr_from_sse, shard_bounds
from egonerf_amd.sampler import SimpleSampler, ThetaImportanceSampler
from tests.helpers import make_model


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    names = _lib.header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.PROTOTYPES)  # the binding covers the whole header, nothing extra
    assert lib.ego_abi_version() == 5
    assert [lib.ego_sizeof(i) for i in range(3)] == [ctypes.sizeof(_lib.Scene), ctypes.sizeof(_lib.RenderArgs),
                                                     ctypes.sizeof(_lib.VmField)]
    assert lib.ego_packed_floats() == 2 * 46852 + 9216  # fp32 layout + fp16-split layout + fp16-table basis fragments


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    assert lib.ego_sample_ray_exp(None, None, None, 0.0, 4, 8, None, None, None) == -1
    assert b"sample_ray_exp" in lib.ego_last_error()
    sc = _lib.Scene()
    assert lib.ego_shade(sc, None, None, None, 1, 1, None, None, None, None) == -1
    assert lib.ego_render_workspace_bytes(4096, None) == -1
    a = _lib.RenderArgs()
    a.n_coarse, a.n_fine, a.resampling, a.use_coarse_sample = 128, 128, 1, 1
    need = lib.ego_render_workspace_bytes(4096, ctypes.byref(a))
    assert need >= 4 * 4096 * (128 * 2 + 256 * 2 + 1 + 256 * 3)


def test_resolution_rule_and_constants(golden):
    fx = golden("stages")
    cfg = synth.SceneConfig()
    c = YinYangSphericalCoords("cpu", cfg.aabb, exp_r=True, N_voxel=cfg.n_voxel, r0=cfg.r0, interval_th=True)
    assert c.resolution == [150, 172, 516]
    assert [c.N_to_reso(n ** 3) for n in (20, 40, 100)] == [[10, 10, 30], [20, 22, 64], [50, 56, 168]]
    assert float(c.far[0]) == float(fx["normr/full/far_r"])
    assert synth.n_to_reso(27e6) == c.resolution


def test_sample_schedules_bit_exact(golden):
    fx = golden("stages")
    for name, (near, far, r0) in dict(indoor=(0.01, 15.0, 0.03), ricoh=(0.1, 300.0, 0.05), mid=(0.01, 50.0, 0.05)).items():
        cfg = synth.SceneConfig(n_voxel=20 ** 3, near=near, far=far, r0=r0)
        c = YinYangSphericalCoords("cpu", cfg.aabb, exp_r=True, N_voxel=cfg.n_voxel, r0=r0, interval_th=True)
        for S in (32, 64, 128, 256, 512):
            z = near + c.sample_schedule(near, far, S)
            assert np.array_equal(z.numpy(), fx[f"sched/{name}/{S}"]), (name, S)


def test_r_lut_reproduces_reference_normalize_r(golden):
    """The host-built LUT + the searchsorted/lerp rule (restated in numpy) gives the reference's normalize_r."""
    fx = golden("stages")
    for name, nv in (("full", 27_000_000), ("tiny", 20 ** 3)):
        cfg = synth.SceneConfig(n_voxel=nv)
        c = YinYangSphericalCoords("cpu", cfg.aabb, exp_r=True, N_voxel=nv, r0=cfg.r0, interval_th=True)
        G = c.reference_r_grid().numpy()
        r = fx[f"normr/{name}/r"]
        k_out = np.clip(np.searchsorted(G, r, side="right"), 1, len(G) - 1)
        k_in = k_out - 1
        out = ((k_in.astype(np.float32) + (r - G[k_in]) / (G[k_out] - G[k_in])) / np.float32(c.N_r)).astype(np.float32)
        assert np.array_equal(out, fx[f"normr/{name}/out"])


def test_samplers_known_answers(golden):
    fx = golden("stages")
    np.random.seed(20221028)  # train.py:413
    s = SimpleSampler(10, 4)
    got = np.stack([s.nextids().numpy() for _ in range(5)])
    assert np.array_equal(got, fx["sampler/simple"])
    assert got.tolist()[0] == [5, 1, 0, 4]
    np.random.seed(20221028)
    t = ThetaImportanceSampler(5, 3, (8, 4), 6, [0, 1, 0, 1])
    assert np.allclose(t.weight, fx["sampler/theta_weight"], rtol=0, atol=1e-15)
    assert np.array_equal(np.asarray(t.nextids()), fx["sampler/theta_ids"])


def test_model_state_dict_matches_reference_manifest(golden):
    fx = golden("tiny")
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    w = synth.make_weights(cfg, seed=1234)
    model = make_model(cfg, w, "cpu")
    sd = model.state_dict()
    ref_keys = sorted(k[len("bw_grad/"):] for k in fx.files if k.startswith("bw_grad/"))  # the reference's named_parameters
    assert sorted(sd) == ref_keys and len(ref_keys) == 32
    for k in ref_keys:
        assert tuple(sd[k].shape) == fx["bw_grad/" + k].shape, k
        assert np.array_equal(sd[k].numpy(), w[k])
    p = model.density_plane_yin[0]
    assert p.shape == (1, 16, 10, 10) and p.permute(0, 2, 3, 1).is_contiguous()  # channel-last memory
    assert model.app_line_yang[2].permute(0, 2, 3, 1).is_contiguous()
    groups = model.get_optparam_groups(0.02, 1e-3, 0.005)
    assert len(groups) == 11 and [g["lr"] for g in groups[:5]] == [0.02] * 4 + [1e-3]


def test_no_cpu_fallback():
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    model = make_model(cfg, synth.make_weights(cfg, seed=1), "cpu")
    for call in (lambda: model(torch.zeros(4, 6), n_coarse=8, exp_sampling=True),
                 lambda: model.compute_densityfeature(torch.zeros(4, 7)),
                 lambda: model.compute_appfeature(torch.zeros(4, 7)),
                 lambda: model.coordinates.from_cartesian(torch.zeros(4, 3)),
                 lambda: model.feature2density(torch.zeros(4))):
        with pytest.raises(RuntimeError, match="HIP device"):
            call()


def test_product_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "egonerf_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_shard_bounds_partition():
    for n, w in ((10, 3), (2048 * 1024, 8), (5, 8), (0, 2)):
        blocks = [shard_bounds(n, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1
    assert abs(psnr_from_sse(0.01 * 300, 300) - 20.0) < 1e-12


def test_reference_checkpoint_loads_on_cpu(golden):
    """A `.th` file written by the reference's own EgoNeRF.save (tests/golden/reference_ckpt.th) unpickles through the
    module shims and rebuilds the model: kwargs objects, state_dict, packed alpha mask, envmap, global_step."""
    import sys
    from egonerf_amd.compat import load_reference_checkpoint
    from egonerf_amd.coordinates import YinYangSphericalCoords
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_ckpt.th")
    model, step = load_reference_checkpoint(path, device="cpu")
    assert step == 4321 and "models.coordinates" not in sys.modules  # shims are removed after the load
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=16)
    w = synth.make_weights(cfg, seed=77)
    for k, v in model.state_dict().items():
        assert np.array_equal(v.numpy(), w[k]), k
    assert isinstance(model.coordinates, YinYangSphericalCoords) and model.coordinates.resolution == [10, 10, 30]
    assert np.array_equal(model.coordinates.reference_r_grid().numpy(),
                          make_model(cfg, w, "cpu").coordinates.reference_r_grid().numpy())
    assert np.array_equal(model.envmap.emission.detach().numpy(), w["envmap.emission"])
    assert model.alphaMask is not None and tuple(model.alphaMask.alpha_volume_yin.shape) == (1, 1, 30, 10, 10)
    assert model.near_far == [0.01, 15.0] and model.density_shift == -8 and model.fea2denseAct == "softplus"


def test_kernels_have_no_high_half_broadcast_packed_fp32_ops(tmp_path):
    """every build of ego_shade.hip whose kernels contained packed fp32 instructions broadcasting the HIGH dword of
    a register pair (`v_pk_fma_f32 ... op_sel:[1,0,0]` without op_sel_hi) returned wrong results in some calls on MI355X; builds
    without them never did.  Compile every source to gfx950 assembly with the build's flags and check that none is there."""
    import os, re, subprocess
    from egonerf_amd import build
    procs = []
    for f in build.SOURCES:
        out = str(tmp_path / f.replace(".hip", ".s"))
        cmd = [build._hipcc(), *[x for x in build.COMMON_FLAGS if x != "-fPIC"], *build.EXTRA_FLAGS.get(f, []), "-S", "--cuda-device-only",
               "-o", out, os.path.join(build.CSRC, f)]
        procs.append((f, out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    n_packed = 0
    for f, out, pr in procs:
        log, _ = pr.communicate()
        assert pr.returncode == 0, (f, log[-2000:])
        lines = [l.strip() for l in open(out) if re.search(r"v_pk_(fma|mul|add)_f32", l)]
        n_packed += len(lines)
        bad = [l for l in lines if "op_sel:" in l and "op_sel_hi" not in l]
        assert not bad, (f, bad[:5])
    assert n_packed > 1000  # the check looked at real code
