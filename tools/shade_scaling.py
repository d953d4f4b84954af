"""Fixed cost of a shade launch: ego_shade_composite at 1024 ... 8192 rays x 512 samples (whole rays per wave), linear fit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egonerf_amd import synth, _lib
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), "cuda")
lib, st, sc = _lib.load(), _lib.stream_handle(), model.scene()
S, dev = 512, "cuda"
res = []
for N in (2048, 4096, 6144, 8192):
    rays = torch.from_numpy(synth.make_rays(N, seed=1)).cuda()
    z = torch.empty(N, S, device=dev); w = torch.empty_like(z); bg = torch.empty(N, device=dev); crd = torch.empty(N, S, 4, device=dev)
    rgb_map = torch.empty(N, 3, device=dev); depth = torch.empty(N, device=dev)
    sched = model._sched(S, dev)
    _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), None, 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march")
    fn = lambda: _lib.check(lib.ego_shade_composite(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), w.data_ptr(), bg.data_ptr(), N, S, None, rgb_map.data_ptr(), depth.data_ptr(), None, None, st), "sc")
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    res.append((N, e0.elapsed_time(e1) / 100))
    print(N, round(res[-1][1], 4), "ms")
n = np.array([r[0] for r in res], float); t = np.array([r[1] for r in res])
b, a = np.polyfit(n, t, 1)
print(f"fit: {a * 1e3:.1f} us fixed + {b * 4096 * 1e3:.1f} us per 4096 rays")
