"""Randomised parity campaign: the HIP render vs the CPU oracle over random scene seeds, grids, ray / sample counts,
resampling modes and envmap settings (eval mode; tolerance of north_star: 1e-4 RGB).  Prints the worst errors."""
import sys, os, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from egonerf_amd.synth import build_model
from oracle.egonerf_oracle import OracleScene

dev = torch.device("cuda", 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
worst = dict(rgb=0.0, depth=0.0, alpha=0.0)
torch.set_num_threads(16)
from tests.helpers import campaign_cases
for case, cfg, w, rays, kw in campaign_cases(int(sys.argv[1]) if len(sys.argv) > 1 else 0, n_cases):
    env, N, resampling = cfg.use_envmap, rays.shape[0], kw["resampling"]
    model, oracle = build_model(cfg, w, dev), OracleScene(cfg, w)
    model.mlp_precision = os.environ.get("EGO_PREC", model.mlp_precision)   # f16f8 (default) | f16x3 | f32
    with torch.no_grad():
        got = model(rays.to(dev), exp_sampling=True, **kw)
        ref = oracle.forward(rays, **kw)
    # A ray whose float32 and float64 REFERENCE evaluations already differ by more than the tolerance is ill-conditioned in the
    # reference itself (sample_pdf: u = 0 / 1 or a cdf knot within rounding, `denom < 1e-5 -> 1`; a sample within an ulp of a yin /
    # yang border): its outcome is decided by the rounding of one sum, so it is reported but not counted
    per_ray = (got[0].cpu() - ref[0]).abs().max(dim=1).values
    bad = torch.nonzero(per_ray > 1e-4).flatten()
    excused = []
    if len(bad):
        o64 = OracleScene(cfg, w, dtype=torch.float64)
        with torch.no_grad():
            r64 = o64.forward(rays[bad].double(), **kw)
        # ... or if the reference's own result moves by more than the tolerance when u moves by the float32 RESOLUTION OF THE CDF
        # (2 ulp of 1 = 2.4e-7): sample_pdf evaluates (u - cdf_lo) / (cdf_hi - cdf_lo) * bin width with a float32 cdf near 1, so a
        # cdf step of 2e-5 (a nearly empty last bin) turns one ulp into 0.6 % of a bin - and the far bins of the exponential schedule
        # are wide.  Which side of such a knot a value lands on depends on the rounding of `weights.sum(-1)` (a float32 cascade sum in
        # ATen, grouped by the CPU's vector width; the library rounds the exact sum once) - tools/pdf_ray_probe.py: seed 4 case 40
        # ray 80 moves by 0.33 = 54.5 (bin) x 1.2e-7 / 1.97e-5 exactly that way.
        nudged = []
        for du in (2.4e-7, -2.4e-7):
            orig = OracleScene.sample_pdf
            def nudge(bins, weights, n, u=None, _du=du):
                if u is None:
                    u = torch.linspace(0.0, 1.0, steps=n, dtype=weights.dtype).expand(list(weights.shape[:-1]) + [n])
                return orig(bins, weights, n, (u.double() + _du).clamp(0.0, 1.0).to(weights.dtype))
            OracleScene.sample_pdf = staticmethod(nudge)
            try:
                with torch.no_grad():
                    nudged.append(oracle.forward(rays[bad], **kw)[0])
            finally:
                OracleScene.sample_pdf = staticmethod(orig)
        for k, b in enumerate(bad.tolist()):
            if float((r64[0][k].float() - ref[0][b]).abs().max()) > 1e-4 or \
                    max(float((nd[k] - ref[0][b]).abs().max()) for nd in nudged) > 1e-4 or \
                    float((got[0][b].cpu().double() - r64[0][k]).abs().max()) <= 1e-5:   # ... or this result IS the float64 one
                excused.append(b)
        keep = torch.ones(len(per_ray), dtype=torch.bool)
        keep[excused] = False
        got = tuple(None if t is None else t.cpu()[keep] for t in got)
        ref = tuple(None if t is None else t[keep] for t in ref)
    e_rgb = float((got[0].cpu() - ref[0]).abs().max()) if len(ref[0]) else 0.0
    e_dep = float((got[1].cpu() - ref[1]).abs().max()) / max(float(ref[1].abs().max()), 1.0)
    e_alpha = float((got[4].cpu() - ref[4]).abs().max()) if not resampling else 0.0
    worst = dict(rgb=max(worst["rgb"], e_rgb), depth=max(worst["depth"], e_dep), alpha=max(worst["alpha"], e_alpha))
    flag = ("" if e_rgb <= 1e-4 else "   <-- ABOVE TOLERANCE") + (f"   [{len(excused)} ray(s) ill-conditioned in the reference (fp32 vs fp64 oracle differ by > 1e-4, the fp32 oracle moves by > 1e-4 with u +- the cdf's float32 resolution, or HIP equals the fp64 oracle to 1e-5): {excused}]" if excused else "")
    print(f"case {case:2d}: grid {cfg.grid} env {int(env)} near/far {cfg.near}/{cfg.far} N {N:3d} {kw}  rgb {e_rgb:.2e} depth(rel) {e_dep:.2e} alpha {e_alpha:.2e}{flag}")
print("worst:", worst)
sys.exit(0 if worst["rgb"] <= 1e-4 else 1)
