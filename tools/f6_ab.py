"""A/B of the MLP arithmetics on one box: stand-alone MLP and fused shade times (alternated), per-sample colour error vs f16x3."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth, _lib
from egonerf_amd.synth import build_model as make_model

dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = make_model(cfg, synth.make_weights(cfg, seed=1234), dev)
N, S = 4096, 512
M = N * S
rays = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
lib, st = _lib.load(), _lib.stream_handle()
z = torch.empty(N, S, device=dev); w = torch.empty_like(z); alpha = torch.empty_like(z); bg = torch.empty(N, device=dev)
crd = torch.empty(N, S, 4, device=dev)
sched = model._sched(S, dev)
sc = model.scene()
_lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), alpha.data_ptr(), 0,
                                 w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march")
feat = torch.empty(M, 27, device=dev)
c7 = torch.zeros(M, 7, device=dev)
flat = crd.view(M, 4); yang = flat[:, 3] != 0
c7[~yang, 0:3] = flat[~yang, 0:3]; c7[yang, 3:6] = flat[yang, 0:3]; c7[:, 6] = flat[:, 3]
_lib.check(lib.ego_app_feature(sc, c7.data_ptr(), M, feat.data_ptr(), st), "app")
vd = rays[:, 3:6].view(N, 1, 3).expand(N, S, 3).contiguous().view(M, 3)

def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

precs = sys.argv[1:] or ["f16f8", "f16f6"]
out = {}
rgbs = {}
for rnd in range(3):
    for prec in ["f16x3"] + precs if rnd == 0 else precs:
        model.mlp_precision = prec
        sc = model.scene()
        rgb = torch.empty(M, 3, device=dev)
        t_mlp = timeit(lambda: _lib.check(lib.ego_mlp_fea(sc, vd.data_ptr(), feat.data_ptr(), M, rgb.data_ptr(), st), "mlp"))
        rgb2 = torch.empty(N, S, 3, device=dev)
        t_sh = timeit(lambda: _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N, S, rgb2.data_ptr(), None, None, st), "shade"))
        out.setdefault(prec, []).append((round(t_mlp, 4), round(t_sh, 4)))
        rgbs[prec] = (rgb, rgb2)
for prec in precs:
    e1 = (rgbs[prec][0] - rgbs["f16x3"][0]).abs(); e2 = (rgbs[prec][1] - rgbs["f16x3"][1]).abs()
    wsum = (e2.view(N, S, 3) * w.view(N, S, 1)).sum(1)
    print(prec, "mlp/shade ms by round:", out[prec], "| per-sample |d rgb| vs f16x3: mlp max %.2e rms %.2e, shade max %.2e | weighted sum per ray max %.2e" %
          (e1.max().item(), e1.pow(2).mean().sqrt().item(), e2.max().item(), wsum.max().item()))
print("f16x3", out["f16x3"])
