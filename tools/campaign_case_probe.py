"""One case of the randomised parity campaign under the microscope: for the worst rays, |HIP - f32 oracle|, |HIP - f64 oracle| and
|f32 oracle - f64 oracle| in all three arithmetics (is a large difference the reference's own float32 conditioning, or ours?).
  python tools/campaign_case_probe.py <seed> <case> [<case> ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import campaign_cases, make_model, make_oracle
torch.set_num_threads(16)
seed, want = int(sys.argv[1]), [int(c) for c in sys.argv[2:]]
for case, cfg, w, rays, kw in campaign_cases(seed, max(want) + 1):
    if case not in want:
        continue
    o32, o64 = make_oracle(cfg, w), make_oracle(cfg, w, dtype=torch.float64)
    with torch.no_grad():
        ref = o32.forward(rays, **kw)
        r64 = o64.forward(rays.double(), **kw)
    # the float64 oracle with the inverse CDF's discontinuity at u = 1 taken the OTHER way: sample_pdf looks up u = 1 (the last entry
    # of linspace(0, 1, n)) with searchsorted(right=True); whether cdf[-1] rounds to <= 1 or > 1 decides between the last bin and the
    # one before it whenever the last bin is thinner than 1e-5 (`denom < 1e-5 -> 1`, ray_utils.py:180-181)
    orig = type(o64).sample_pdf
    def flipped(bins, weights, n, u=None):
        w_ = weights + 1e-5
        pdf = w_ / w_.sum(-1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
        side = cdf[..., -1:] > 1
        cdf = cdf.clone(); cdf[..., -1:] = torch.where(side, torch.ones_like(side, dtype=cdf.dtype), torch.full_like(cdf[..., -1:], 1 + 1e-12))
        if u is None:
            u = torch.linspace(0.0, 1.0, steps=n, dtype=cdf.dtype).expand(list(cdf.shape[:-1]) + [n])
        u = u.contiguous()
        idx = torch.searchsorted(cdf.detach(), u, right=True)
        lo = (idx - 1).clamp(min=0); hi = idx.clamp(max=cdf.shape[-1] - 1)
        c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
        b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
        den = c_hi - c_lo
        den = torch.where(den < 1e-5, torch.ones_like(den), den)
        return b_lo + (u - c_lo) / den * (b_hi - b_lo)
    type(o64).sample_pdf = staticmethod(flipped)
    try:
        with torch.no_grad():
            r64_flip = o64.forward(rays.double(), **kw)
    finally:
        type(o64).sample_pdf = staticmethod(orig)
    print(f"seed {seed} case {case}: grid {cfg.grid} near/far {cfg.near}/{cfg.far} shift {cfg.density_shift} N {rays.shape[0]} {kw}")
    for prec in ("f16f8", "f16x3", "f32"):
        model = make_model(cfg, w, "cuda")
        model.mlp_precision = prec
        with torch.no_grad():
            got = model(rays.cuda(), exp_sampling=True, **kw)
        rgb = got[0].cpu()
        per = (rgb - ref[0]).abs().max(1).values
        for b in torch.argsort(per, descending=True)[:3].tolist():
            print(f"  {prec} ray {b}: |HIP-f32| {float(per[b]):.2e} |HIP-f64| {float((rgb[b].double() - r64[0][b]).abs().max()):.2e} "
                  f"|f32-f64| {float((ref[0][b].double() - r64[0][b]).abs().max()):.2e} |HIP-f64 with the u=1 sample on the other side| "
                  f"{float((rgb[b].double() - r64_flip[0][b]).abs().max()):.2e}  depth hip {got[1][b].item():.6f} f32 {ref[1][b].item():.6f} "
                  f"f64 {r64[1][b].item():.6f} f64-flipped {r64_flip[1][b].item():.6f}")
