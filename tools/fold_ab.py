"""A/B inside one process: EgoNeRF.forward at the headline shape with compositing folded into the shade kernel (EGO_RENDER_FOLD=1, read by the
library on every call) and with the two-launch form (default), alternated."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egonerf_amd import synth
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), "cuda")
rays = torch.from_numpy(synth.make_rays(4096, seed=1)).cuda()
kw = dict(n_coarse=512, exp_sampling=True)
def run(steps=200):
    with torch.no_grad():
        for _ in range(20): model(rays, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): model(rays, **kw)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
run(400)
for rnd in range(4):
    os.environ["EGO_RENDER_FOLD"] = "1"; a = run()
    os.environ.pop("EGO_RENDER_FOLD", None); b = run()
    print(f"round {rnd}: folded {a:.4f} ms/step, two launches {b:.4f} ms/step")
# the two shade kernels alone
from egonerf_amd import _lib
lib, st, sc = _lib.load(), _lib.stream_handle(), model.scene()
N, S = 4096, 512
dev = "cuda"
z = torch.empty(N, S, device=dev); w = torch.empty_like(z); bg = torch.empty(N, device=dev); crd = torch.empty(N, S, 4, device=dev)
rgb = torch.empty(N, S, 3, device=dev); rgb_map = torch.empty(N, 3, device=dev); depth = torch.empty(N, device=dev)
sched = model._sched(S, dev)
_lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), None, 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march")
def ev_time(fn, reps=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(3):
    a = ev_time(lambda: _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N, S, rgb.data_ptr(), None, None, st), "shade"))
    b = ev_time(lambda: _lib.check(lib.ego_shade_composite(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), w.data_ptr(), bg.data_ptr(), N, S, None, rgb_map.data_ptr(), depth.data_ptr(), None, None, st), "sc"))
    c = ev_time(lambda: _lib.check(lib.ego_composite(sc, rays.data_ptr(), z.data_ptr(), w.data_ptr(), bg.data_ptr(), rgb.data_ptr(), N, S, rgb_map.data_ptr(), depth.data_ptr(), None, None, None, st), "comp"))
    print(f"kernels: ego_shade {a:.4f} ms, ego_shade_composite {b:.4f} ms, ego_composite {c:.4f} ms")
