"""Times the phases of the shade kernel in isolation at the bench size (M = 4096 x 512 samples):
gather+basis only (ego_app_feature), MLP only (ego_mlp_fea), and the fused kernel (ego_shade)."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth, _lib
from egonerf_amd.synth import build_model as make_model

dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = make_model(cfg, synth.make_weights(cfg, seed=1234), dev)
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
model.mlp_precision = prec
if len(sys.argv) > 2:
    model.app_table_dtype = sys.argv[2]  # "f16": half-precision shadow of the appearance tables
N, S = 4096, 512
M = N * S
rays = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
lib, st, sc = _lib.load(), _lib.stream_handle(), model.scene()
z = torch.empty(N, S, device=dev); w = torch.empty_like(z); alpha = torch.empty_like(z); bg = torch.empty(N, device=dev)
crd = torch.empty(N, S, 4, device=dev); rgb = torch.empty(N, S, 3, device=dev)
sched = model._sched(S, dev)
_lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), alpha.data_ptr(), 0,
                                 w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march")
c7 = torch.zeros(M, 7, device=dev)
flat = crd.view(M, 4)
yang = flat[:, 3] != 0
c7[~yang, 0:3] = flat[~yang, 0:3]; c7[yang, 3:6] = flat[yang, 0:3]; c7[:, 6] = flat[:, 3]
feat = torch.empty(M, 27, device=dev)
vd = rays[:, 3:6].view(N, 1, 3).expand(N, S, 3).contiguous().view(M, 3)

def timeit(fn, reps=10):
    for _ in range(2): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))

res = dict(precision=prec)
res["app_only_ms"] = timeit(lambda: _lib.check(lib.ego_app_feature(sc, c7.data_ptr(), M, feat.data_ptr(), st), "app"))
res["mlp_only_ms"] = timeit(lambda: _lib.check(lib.ego_mlp_fea(sc, vd.data_ptr(), feat.data_ptr(), M, rgb.data_ptr(), st), "mlp"))
res["shade_ms"] = timeit(lambda: _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N, S, rgb.data_ptr(), None, None, st), "shade"))
res["march_ms"] = timeit(lambda: _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(),
                                                                   alpha.data_ptr(), 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march"))
# same-texel variant: every sample at the same coordinates (all gathers hit one texel set -> pure L1 hits)
c7s = c7[:1].expand(M, 7).contiguous()
res["app_only_same_texel_ms"] = timeit(lambda: _lib.check(lib.ego_app_feature(sc, c7s.data_ptr(), M, feat.data_ptr(), st), "app"))
print(json.dumps(res))
