"""Randomised parity campaign inside the GPU suite: HIP render vs the CPU oracle over random scene seeds,
tiny grids, ray / sample counts, resampling modes and envmap settings, in every shipped split arithmetic - the streams of seeds 13 and 23
(whose cases 6 / 7 were the worst of round 2's 160-case campaign) and 4 and 7 (seed 4: the worst of round 3), 40 cases each, and the full
4 x 160-case campaign of seeds 0-3, which leaves its summary in gpurun_out/parity_campaign_full.json -> profiles/rNN/.  Original note on seeds 13 / 23: their cases 6 / 7 were the worst of round 2's campaign (|dRGB| 8e-5, from the float32 sensitivity of sample_pdf on
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


PRECS = ("f16f6", "f16f8", "f16x3")   # every shipped split arithmetic (f16f8 had lost its campaign)


def run_campaign(seed: int, n_cases: int):
    """One stream of cases; the float32 oracle (the slow part: CPU) is evaluated ONCE per case and every arithmetic is held against it.
    -> {prec: summary dict}; raises on the first violated bound."""
    torch.set_num_threads(16)
    S = {p: dict(worst=0.0, watched=0, excused=0, ratio=0.0, worst_dpsnr=0.0, worst_case=None) for p in PRECS}
    for case, cfg, w, rays, kw in campaign_cases(seed, n_cases):
        model, oracle = make_model(cfg, w, "cuda"), make_oracle(cfg, w)
        with torch.no_grad():
            ref = oracle.forward(rays, **kw)
        o64_cache = {}

        def f64_rows(look):
            key = tuple(look.tolist())
            if key not in o64_cache:
                o64 = make_oracle(cfg, w, dtype=torch.float64)
                with torch.no_grad():
                    o64_cache[key] = o64.forward(rays[look].double(), **kw)[0]
            return o64_cache[key]

        for prec in PRECS:
            st = S[prec]
            model.mlp_precision = prec
            with torch.no_grad():
                got = model(rays.cuda(), exp_sampling=True, **kw)
            rgb = got[0].cpu()
            if rays.shape[0] >= 64:   # north_star's PSNR clause on a ~30 dB target (a handful of rays is not an image: one ray moves its PSNR)
                d_psnr = synth.delta_psnr(rgb.numpy(), ref[0].numpy(), seed=seed * 100 + case)[0]
                assert abs(d_psnr) <= 1e-3, (seed, case, prec, d_psnr)
                st["worst_dpsnr"] = max(st["worst_dpsnr"], abs(d_psnr))
            per_ray = (rgb - ref[0]).abs().max(dim=1).values
            look = torch.nonzero(per_ray > WATCH).flatten()
            if len(look):
                r64 = f64_rows(look)
                for k, b in enumerate(look.tolist()):
                    d_hip = float((rgb[b].double() - r64[k]).abs().max())
                    d_f32 = float((ref[0][b].double() - r64[k]).abs().max())
                    where = f"seed {seed} case {case} ray {b} ({prec}; grid {cfg.grid}, {kw})"
                    assert d_hip <= max(K * d_f32, WATCH), f"{where}: |HIP - f64| = {d_hip:.2e} but |f32 oracle - f64| = {d_f32:.2e}"
                    st["watched"] += 1
                    st["ratio"] = max(st["ratio"], d_hip / max(d_f32, 1e-12)) if d_hip > WATCH else st["ratio"]
                    if per_ray[b] > TOL:
                        # ... unless this result IS the float64 one to a tenth of the tolerance: then |HIP - f32 oracle| <= the float32
                        # oracle's own error + 1e-5 (seed 4 case 22 ray 243: |HIP - f64| 5e-7, |f32 oracle - f64| 9.95e-5)
                        assert d_f32 > TOL or d_hip <= 0.1 * TOL, f"{where}: |HIP - f32 oracle| = {float(per_ray[b]):.2e} on a well-conditioned ray"
                        st["excused"] += 1
                        per_ray[b] = 0.0
            if float(per_ray.max()) > st["worst"]:
                st["worst"], st["worst_case"] = float(per_ray.max()), case
            # depth and (without resampling) per-sample alpha ride along, as in tools/parity_campaign.py
            keep = per_ray > -1
            if len(look):
                keep[look] = False
            if bool(keep.any()):
                e_dep = float((got[1].cpu()[keep] - ref[1][keep]).abs().max()) / max(float(ref[1].abs().max()), 1.0)
                assert e_dep <= 1e-3, (seed, case, prec, e_dep)
                if not kw["resampling"]:
                    assert float((got[4].cpu()[keep] - ref[4][keep]).abs().max()) <= 1e-4, (seed, case, prec)
    for prec, st in S.items():
        assert st["worst"] <= TOL, (seed, prec, st)
        print(f"campaign seed {seed} x {n_cases} {prec}: worst |dRGB| {st['worst']:.2e} (case {st['worst_case']}), {st['watched']} rays above {WATCH:g} checked against "
              f"float64, {st['excused']} ill-conditioned in the reference, worst |HIP - f64| / |f32 oracle - f64| = {st['ratio']:.2f}, "
              f"worst |delta PSNR| on a 30 dB target {st['worst_dpsnr']:.1e} dB")
    return S


@pytest.mark.parametrize("seed", [13, 23, 4, 7])
def test_campaign_vs_float32_and_float64_oracle(seed):
    S = run_campaign(seed, 40)
    for prec, st in S.items():
        assert st["excused"] <= MAX_EXCUSED, f"seed {seed} {prec}: {st['excused']} rays needed the float64 excuse (allowed: {MAX_EXCUSED})"


def test_full_campaign_4_seeds_x_160_cases_x_3_arithmetics():
    """the whole campaign (seeds 0-3 x 160 cases x f16f6 / f16f8 / f16x3) - what used to live in tools/soak_*.sh - is
    part of the plain `-m gpu` suite: with the float32 oracle evaluated once per case it takes 15 s on the GPU box.  Same bounds as the
    per-seed test above, except that the excused-ray allowance scales with the case count (MAX_EXCUSED per 40 cases).  The summary
    (worst |dRGB|, excused rays, worst dPSNR per seed and arithmetic) is written to gpurun_out/parity_campaign_full.json (copied to
    profiles/rNN/ at the end of a round).  (tests/conftest.py keeps a `slow` marker for soaks that do not fit the suite: none today.)"""
    import json
    import os
    from egonerf_amd.build import source_hash
    out = dict(what="tests/test_hip_parity_campaign.py::test_full_campaign_4_seeds_x_160_cases_x_3_arithmetics", source_hash=source_hash(),
               tolerance_rgb=TOL, watch=WATCH, K=K, seeds={})
    for seed in (0, 1, 2, 3):
        S = run_campaign(seed, 160)
        out["seeds"][str(seed)] = S
        for prec, st in S.items():
            assert st["excused"] <= 4 * MAX_EXCUSED, (seed, prec, st)
    out["worst_rgb"] = max(st["worst"] for S in out["seeds"].values() for st in S.values())
    out["excused_total"] = sum(st["excused"] for S in out["seeds"].values() for st in S.values())
    out["worst_delta_psnr_db"] = max(st["worst_dpsnr"] for S in out["seeds"].values() for st in S.values())
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_campaign_full.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("worst_rgb", "excused_total", "worst_delta_psnr_db")}))
