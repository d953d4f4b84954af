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

    default_prec = model.mlp_precision

    def appf(prec, tab, pts):
        def f():
            model.mlp_precision, model.app_table_dtype = prec, tab
            r = model.compute_appfeature(pts)
            model.mlp_precision, model.app_table_dtype = default_prec, "f32"
            return r
        return f

    def withprec(prec, fn):
        def f():
            model.mlp_precision = prec
            try:
                return fn()
            finally:
                model.mlp_precision = default_prec
        return f

    default_prec = model.mlp_precision
    feat = model.compute_appfeature(q_in)
    cases = [("appfeature scattered", appf("f16x3", "f32", q)), ("appfeature in-range", appf("f16x3", "f32", q_in)),
             ("appfeature one grid", appf("f16x3", "f32", q_one)), ("appfeature f32-MFMA", appf("f32", "f32", q)),
             ("appfeature f16 tables", appf("f16x3", "f16", q)), ("densityfeature", lambda: model.compute_densityfeature(q)),
             ("renderModule f16x3", withprec("f16x3", lambda: model.renderModule(q_in, vd, feat))),
             ("renderModule f16f8", withprec("f16f8", lambda: model.renderModule(q_in, vd, feat))),
             ("renderModule f16f6", withprec("f16f6", lambda: model.renderModule(q_in, vd, feat))),
             ("forward 24 f16f6", withprec("f16f6", lambda: model(rays, n_coarse=24, exp_sampling=True)[0])),
             ("forward 32 f16f6", withprec("f16f6", lambda: model(rays, n_coarse=32, exp_sampling=True)[0])),
             ("forward 16+16 f16f6", withprec("f16f6", lambda: model(rays, n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)[0])),
             ("forward 24 f16x3", withprec("f16x3", lambda: model(rays, n_coarse=24, exp_sampling=True)[0])),
             ("forward 24 f16f8", withprec("f16f8", lambda: model(rays, n_coarse=24, exp_sampling=True)[0])),
             ("forward 16+16 f16x3", withprec("f16x3", lambda: model(rays, n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)[0])),
             ("forward 16+16 f16f8", withprec("f16f8", lambda: model(rays, n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)[0]))]
    if full:
        cfg2 = synth.SceneConfig()
        big = build_model(cfg2, synth.make_weights(cfg2, seed=1234), dev)
        rays2 = torch.from_numpy(synth.make_rays(4096, seed=1)).to(dev)
        for prec in ("f16x3", "f16f8", "f16f6"):
            def big_fn(kw, prec=prec):
                def f():
                    big.mlp_precision = prec
                    return big(rays2, exp_sampling=True, **kw)[0]
                return f
            cases.append((f"forward 4096 x 512 {prec}", big_fn(dict(n_coarse=512))))
            cases.append((f"forward 4096 x (128+128) {prec}", big_fn(dict(n_coarse=128, n_fine=128, resampling=True))))
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
