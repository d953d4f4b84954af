"""Per-layer sweep of the fp16-split arithmetic of the shade kernel, emulated on the CPU in float64.

Every matrix product of the kernel (basis 144->27, layer 1 150->128, layer 2 128->128) is w_hi*x_hi + w_lo*x_hi + w_hi*x_lo with
fp16 operands and fp32 accumulation.  For each of the two correction terms of each layer this tool tries {fp16 (today), fp8 e4m3 with
a fixed 2^11 block scale (v_mfma_scale_f32_32x32x64_f8f6f4: half the matrix-pipe time of an fp16 term), dropped} on the bench scene
(full barbershop grid, real gathered appearance products and compositing weights) and reports the composited max |d RGB| per ray
against float64 next to the matrix-pipe and kernel time the variant would save (additive issue model of bench.py: MFMA 32 clk,
VALU 4 clk; an fp8 term also needs its operands converted: +1 VALU per 4 values).

    python tools/precision_sweep.py [n_rays]     -> table on stdout + profiles/r02/precision_sweep.json
"""
import itertools, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from oracle.egonerf_oracle import OracleScene

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 96
S = 512


def f16(x):
    return x.astype(np.float16).astype(np.float64)


def e4m3(x):
    """round to nearest fp8 e4m3 (bias 7, max 448, subnormals down to 2^-9), saturating"""
    x = np.asarray(x, np.float64)
    s, a = np.sign(x), np.minimum(np.abs(x), 448.0)
    e = np.maximum(np.floor(np.log2(np.maximum(a, 2.0 ** -30))), -6.0)
    q = 2.0 ** (e - 3)
    return s * np.round(a / q) * q


def product(W, X, wl, xl):
    """X [n,k] @ W[m,k]^T as the kernel would, correction terms per `wl` (w_lo * x_hi) / `xl` (w_hi * x_lo) in {f16, fp8, drop}"""
    Wh, Xh = f16(W), f16(X)
    out = Xh @ Wh.T
    if wl == "f16":
        out = out + Xh @ f16(W - Wh).T
    elif wl == "fp8":
        out = out + (e4m3(Xh) @ e4m3((W - Wh) * 2.0 ** 11).T) / 2.0 ** 11
    if xl == "f16":
        out = out + f16(X - Xh) @ Wh.T
    elif xl == "fp8":
        out = out + (e4m3((X - Xh) * 2.0 ** 11) @ e4m3(Wh).T) / 2.0 ** 11
    return out


def pe(v):
    p = (v[..., None] * np.array([1.0, 2.0])).reshape(v.shape[0], -1)
    return np.concatenate([np.sin(p), np.cos(p)], 1)


cfg = synth.SceneConfig()
w = synth.make_weights(cfg, seed=1234)
sc = OracleScene(cfg, w, dtype=torch.float64)
rays = torch.from_numpy(synth.make_rays(n_rays, seed=1))
with torch.no_grad():
    (_, inter) = sc.forward(rays, n_coarse=S, keep=True)
    c7n = inter["c7n"].reshape(-1, 7)
    # the 144 gathered products per sample (what feeds the basis MFMAs), per grid
    V = np.zeros((c7n.shape[0], 144))
    is_yin = (c7n[:, 6] == 0).numpy()
    for g, sel, base in (("yin", is_yin, 0), ("yang", ~is_yin, 3)):
        p3 = c7n[torch.from_numpy(sel)][:, base:base + 3]
        taps = sc._vm_taps([sc.table("app", "plane", g, i) for i in range(3)], [sc.table("app", "line", g, i) for i in range(3)], p3)
        V[sel] = (torch.cat([t[0] for t in taps]) * torch.cat([t[1] for t in taps])).T.numpy()
weight = inter["weight"].numpy()
dirs = np.repeat(rays[:, 3:6].double().numpy(), S, 0)
B = {g: w[f"basis_mat_{g}.weight"].astype(np.float64) for g in ("yin", "yang")}
W1, b1, W2, b2, W3, b3 = (np.asarray(w[k], np.float64) for k in ("renderModule.mlp.0.weight", "renderModule.mlp.0.bias", "renderModule.mlp.2.weight",
                                                                   "renderModule.mlp.2.bias", "renderModule.mlp.4.weight", "renderModule.mlp.4.bias"))


def render(modes):
    """modes = dict(layer -> (wl, xl)) or None for float64"""
    fe = np.zeros((V.shape[0], 27))
    for g, sel in (("yin", is_yin), ("yang", ~is_yin)):
        fe[sel] = V[sel] @ B[g].T if modes is None else product(B[g], V[sel].astype(np.float32).astype(np.float64), *modes["basis"])
    if modes is not None:
        fe = fe.astype(np.float32).astype(np.float64)
    x = np.concatenate([fe, dirs, pe(fe), pe(dirs)], 1)
    h1 = np.maximum((x @ W1.T if modes is None else product(W1, x.astype(np.float32).astype(np.float64), *modes["l1"])) + b1, 0)
    h2 = np.maximum((h1 @ W2.T if modes is None else product(W2, h1.astype(np.float32).astype(np.float64), *modes["l2"])) + b2, 0)
    rgb = 1 / (1 + np.exp(-(h2 @ W3.T + b3)))   # layer 3 runs in fp32 on the VALU
    return rgb, (weight[..., None] * rgb.reshape(n_rays, S, 3)).sum(1)


ref_s, ref_c = render(None)
K_STEPS = dict(basis=9 * 1, l1=10 * 4, l2=8 * 4)          # MFMAs of one term per 32-sample tile
VALUES = dict(basis=144, l1=160, l2=128)                   # activation values per sample whose x_lo feeds a term
MFMA_NOW, VALU_NOW = 243, 1818                             # profiles/r02/pmc_traffic.json
TOTAL_CLK = MFMA_NOW * 32 + VALU_NOW * 4
rows = []
opts = ["f16", "fp8", "drop"]
for combo in itertools.product(opts, repeat=6):
    modes = dict(basis=combo[0:2], l1=combo[2:4], l2=combo[4:6])
    if sum(c != "f16" for c in combo) > 3 and "drop" in combo:   # keep the sweep small: at most 3 changed terms when dropping
        continue
    d_mfma = d_valu = 0.0
    for layer, (wl, xl) in modes.items():
        for term, m in (("wl", wl), ("xl", xl)):
            if m == "fp8":
                d_mfma -= 0.5 * K_STEPS[layer]
                d_valu += 32 * VALUES[layer] / 64 / 4 * (2 if term == "xl" else 1)   # convert x_hi (and x_lo * 2^11) to fp8: 4 values per op
            elif m == "drop":
                d_mfma -= K_STEPS[layer]
                if term == "xl":
                    d_valu -= 32 * VALUES[layer] / 64 * 1.5                          # no fma_mix residual, no second cvt_pkrtz
    s, c = render(modes)
    rows.append(dict(modes={k: list(v) for k, v in modes.items()}, max_rgb_err_composited=float(np.abs(c - ref_c).max()),
                     max_rgb_err_per_sample=float(np.abs(s - ref_s).max()), mfma_per_tile=MFMA_NOW + d_mfma, valu_per_tile=VALU_NOW + d_valu,
                     est_kernel_time_ratio=((MFMA_NOW + d_mfma) * 32 + (VALU_NOW + d_valu) * 4) / TOTAL_CLK))
rows.sort(key=lambda r: r["est_kernel_time_ratio"])
base = [r for r in rows if all(v == ["f16", "f16"] for v in r["modes"].values())][0]
ok = [r for r in rows if r["max_rgb_err_composited"] <= 1e-5]
print(f"{len(rows)} variants on {n_rays} rays x {S} samples; today (all fp16): composited max |d RGB| {base['max_rgb_err_composited']:.2e}")
print("fastest variants that keep the composited error <= 1e-5 (10x margin to the 1e-4 bar):")
for r in ok[:8]:
    print(f"  time x{r['est_kernel_time_ratio']:.3f}  err {r['max_rgb_err_composited']:.2e}  {r['modes']}")
print("fastest variants overall:")
for r in rows[:6]:
    print(f"  time x{r['est_kernel_time_ratio']:.3f}  err {r['max_rgb_err_composited']:.2e}  {r['modes']}")
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02", "precision_sweep.json")
json.dump(dict(n_rays=n_rays, samples=S, scene="bench scene (barbershop grid, seed 1234)", model="additive issue model: 32 clk per MFMA, 4 clk per VALU",
               today=base, best_within_1e_5=ok[:12], fastest=rows[:12], n_variants=len(rows)), open(out, "w"), indent=1)
