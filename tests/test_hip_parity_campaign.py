"""Randomised parity campaign inside the GPU suite: HIP render vs the CPU oracle over random scene seeds,
tiny grids, ray / sample counts, resampling modes and envmap settings, in both fp16 arithmetics - the streams of seeds 13 and 23,
whose cases 6 / 7 were the worst of round 2's 160-case campaign (|dRGB| 8e-5, from the float32 sensitivity of sample_pdf on
steep tiny grids, dataLoader/ray_utils.py:156-187).

Tolerance (north_star): 1e-4 RGB against the reference's float32 evaluation.  On top of that the argument "what is left comes
from the reference's own fp32 conditioning, not from this implementation" is CHECKED rather than stated: for every ray whose
HIP result is more than 5e-5 from the float32 oracle, the float64 oracle is evaluated and the HIP result must be no farther
from that truth than K = 3 x the float32 oracle is (or within 5e-5 of it; two float32 evaluation orders of an ill-conditioned
inverse CDF have errors of the same order, not of the same size: the worst ratio seen is 2.4, seed 13 case 6 ray 40).  A ray farther than 1e-4 from the float32 oracle is
accepted only if the float32 oracle itself is that far from the float64 one (a discontinuity of the reference algorithm decided
by one rounding: `denom < 1e-5 -> 1`, searchsorted ties, a sample within an ulp of a yin / yang border) and the same K bound holds,
or if the HIP result equals the float64 one to a tenth of the tolerance.  (Since alpha is evaluated as -expm1(-sigma * dist) the
coarse weights carry no cancellation error of their own and the HIP result tracks the float64 oracle to ~1e-6 on such rays: what
is left against the float32 oracle is that oracle's own rounding.)
"""
import numpy as np
import pytest
import torch

from egonerf_amd import synth
from tests.helpers import campaign_cases, make_model, make_oracle

pytestmark = pytest.mark.gpu

TOL, WATCH, K = 1e-4, 5e-5, 3.0
MAX_EXCUSED = 2   # rays per (seed, arithmetic) that may exceed 1e-4 vs the float32 oracle under the float64 argument above; asserted, not printed


@pytest.mark.parametrize("prec", ["f16f6", "f16f8", "f16x3"])   # every shipped split arithmetic (f16f8 had lost its campaign)
@pytest.mark.parametrize("seed", [13, 23])
def test_campaign_vs_float32_and_float64_oracle(seed, prec):
    torch.set_num_threads(16)
    worst, watched, excused, ratio, worst_dpsnr = 0.0, 0, 0, 0.0, 0.0
    for case, cfg, w, rays, kw in campaign_cases(seed, 20):
        model, oracle = make_model(cfg, w, "cuda"), make_oracle(cfg, w)
        model.mlp_precision = prec
        with torch.no_grad():
            got = model(rays.cuda(), exp_sampling=True, **kw)
            ref = oracle.forward(rays, **kw)
        rgb = got[0].cpu()
        if rays.shape[0] >= 64:   # north_star's PSNR clause on a ~30 dB target (a handful of rays is not an image: one ray moves its PSNR)
            d_psnr = synth.delta_psnr(rgb.numpy(), ref[0].numpy(), seed=seed * 100 + case)[0]
            assert abs(d_psnr) <= 1e-3, (seed, case, prec, d_psnr)
            worst_dpsnr = max(worst_dpsnr, abs(d_psnr))
        per_ray = (rgb - ref[0]).abs().max(dim=1).values
        look = torch.nonzero(per_ray > WATCH).flatten()
        if len(look):
            o64 = make_oracle(cfg, w, dtype=torch.float64)
            with torch.no_grad():
                r64 = o64.forward(rays[look].double(), **kw)[0]
            for k, b in enumerate(look.tolist()):
                d_hip = float((rgb[b].double() - r64[k]).abs().max())
                d_f32 = float((ref[0][b].double() - r64[k]).abs().max())
                where = f"seed {seed} case {case} ray {b} ({prec}; grid {cfg.grid}, {kw})"
                assert d_hip <= max(K * d_f32, WATCH), f"{where}: |HIP - f64| = {d_hip:.2e} but |f32 oracle - f64| = {d_f32:.2e}"
                watched += 1
                ratio = max(ratio, d_hip / max(d_f32, 1e-12)) if d_hip > WATCH else ratio
                if per_ray[b] > TOL:
                    # ... unless this result IS the float64 one to a tenth of the tolerance: then |HIP - f32 oracle| <= the float32
                    # oracle's own error + 1e-5 (seed 4 case 22 ray 243: |HIP - f64| 5e-7, |f32 oracle - f64| 9.95e-5)
                    assert d_f32 > TOL or d_hip <= 0.1 * TOL, f"{where}: |HIP - f32 oracle| = {float(per_ray[b]):.2e} on a well-conditioned ray"
                    excused += 1
                    per_ray[b] = 0.0
        worst = max(worst, float(per_ray.max()))
        # depth and (without resampling) per-sample alpha ride along, as in tools/parity_campaign.py
        keep = per_ray > -1
        if len(look):
            keep[look] = False
        if bool(keep.any()):
            e_dep = float((got[1].cpu()[keep] - ref[1][keep]).abs().max()) / max(float(ref[1].abs().max()), 1.0)
            assert e_dep <= 1e-3, (seed, case, e_dep)
            if not kw["resampling"]:
                assert float((got[4].cpu()[keep] - ref[4][keep]).abs().max()) <= 1e-4, (seed, case)
    assert worst <= TOL
    assert excused <= MAX_EXCUSED, f"seed {seed} {prec}: {excused} rays needed the float64 excuse (allowed: {MAX_EXCUSED})"
    print(f"campaign seed {seed} {prec}: worst |dRGB| {worst:.2e}, {watched} rays above {WATCH:g} checked against float64, {excused} ill-conditioned in the reference, "
          f"worst |HIP - f64| / |f32 oracle - f64| = {ratio:.2f}, worst |delta PSNR| on a 30 dB target {worst_dpsnr:.1e} dB")
