"""Stress: the per-point ops and the fused render must be bit-reproducible call to call (there are no atomics on the forward
path).  Catches hardware-hazard / race bugs that a parity test only hits once in a few runs.
  python tools/determinism_check.py [reps] [--full]      (--full adds the 4096 x 512 bench batch)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from egonerf_amd.synth import build_model


def run(reps=200, full=False, verbose=True):
    dev = torch.device("cuda", 0)
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    model = build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
    u = torch.from_numpy(synth.hash_uniform(99, 0, 512 * 7).reshape(512, 7).astype(np.float32))
    q = u * 2.6 - 1.3                      # scattered points, out-of-range coordinates included, both grids in every wave
    q[:, 6] = (u[:, 6] > 0.5).float()
    q = q.to(dev)
    q_in = q.clone(); q_in[:, :6] = q_in[:, :6].clamp(-0.999, 0.999)
    q_one = q_in.clone(); q_one[:, 6] = 0
    rays = torch.from_numpy(synth.make_rays(256, seed=7)).to(dev)
    vd = rays[:, 3:6].repeat(2, 1).contiguous()

    def appf(prec, tab, pts):
        def f():
            model.mlp_precision, model.app_table_dtype = prec, tab
            r = model.compute_appfeature(pts)
            model.mlp_precision, model.app_table_dtype = "f16x3", "f32"
            return r
        return f

    feat = model.compute_appfeature(q_in)
    cases = [("appfeature scattered", appf("f16x3", "f32", q)), ("appfeature in-range", appf("f16x3", "f32", q_in)),
             ("appfeature one grid", appf("f16x3", "f32", q_one)), ("appfeature f32-MFMA", appf("f32", "f32", q)),
             ("appfeature f16 tables", appf("f16x3", "f16", q)), ("densityfeature", lambda: model.compute_densityfeature(q)),
             ("renderModule", lambda: model.renderModule(q_in, vd, feat)),
             ("forward 24", lambda: model(rays, n_coarse=24, exp_sampling=True)[0]),
             ("forward 16+16", lambda: model(rays, n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)[0])]
    if full:
        cfg2 = synth.SceneConfig()
        big = build_model(cfg2, synth.make_weights(cfg2, seed=1234), dev)
        rays2 = torch.from_numpy(synth.make_rays(4096, seed=1)).to(dev)
        cases.append(("forward 4096 x 512 (rgb)", lambda: big(rays2, n_coarse=512, exp_sampling=True)[0]))
        cases.append(("forward 4096 x (128+128)", lambda: big(rays2, n_coarse=128, n_fine=128, exp_sampling=True, resampling=True)[0]))
    out = {}
    with torch.no_grad():
        for name, fn in cases:
            n = max(reps // 10, 10) if "4096" in name else reps
            ref = fn().clone()
            bad, worst = 0, 0.0
            for _ in range(n):
                o = fn()
                if not torch.equal(o, ref):
                    bad += 1
                    worst = max(worst, float((o - ref).abs().max()))
            out[name] = (bad, n, worst)
            if verbose:
                print(f"{name:28s} mismatching calls: {bad} / {n}   worst |diff| {worst:.3g}")
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    res = run(int(args[0]) if args else 200, "--full" in sys.argv)
    sys.exit(1 if any(b for b, _, _ in res.values()) else 0)
