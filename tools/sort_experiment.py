"""Experiment: does ordering the ray batch by direction (L2 locality per XCD) change the shade / march kernel time?"""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth, _lib
from egonerf_amd.synth import build_model as make_model
dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = make_model(cfg, synth.make_weights(cfg, seed=1234), dev)
N, S = 4096, 512
rays0 = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
lib, st, sc = _lib.load(), _lib.stream_handle(), model.scene()
sched = model._sched(S, dev)
def run(rays, reps=10):
    z = torch.empty(N, S, device=dev); w = torch.empty_like(z); alpha = torch.empty_like(z); bg = torch.empty(N, device=dev)
    crd = torch.empty(N, S, 4, device=dev); rgb = torch.empty(N, S, 3, device=dev)
    def once():
        _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), alpha.data_ptr(), 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "m")
        e1.record()
        _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N, S, rgb.data_ptr(), None, None, st), "s")
    tm, ts = [], []
    for i in range(reps + 2):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); once(); e2.record(); torch.cuda.synchronize()
        if i >= 2: tm.append(e0.elapsed_time(e1)); ts.append(e1.elapsed_time(e2))
    return float(np.mean(tm)), float(np.mean(ts))
res = {"random": run(rays0)}
d = rays0[:, 3:6]
theta = torch.acos(d[:, 2].clamp(-1, 1)); phi = torch.atan2(d[:, 1], d[:, 0])
for name, key in (("sorted_theta_phi", (theta * 8 / np.pi).floor() * 1000 + phi), ("sorted_phi", phi), ("sorted_octant", (d[:, 0] > 0).float() * 4 + (d[:, 1] > 0).float() * 2 + (d[:, 2] > 0).float())):
    res[name] = run(rays0[torch.argsort(key)].contiguous())
# XCD-interleaved: block b runs on XCD b % 8 and takes ray ~ b/2 of each 128-ray sweep -> give XCD x the x-th angular octile
order = torch.argsort(phi)
groups = order.view(8, N // 8)              # 8 angular groups of 512 rays
inter = torch.empty_like(order)
# within each sweep of 128 rays, ray slot q (0..127) is served by blocks 2q, 2q+1 -> XCD (2q) % 8, (2q+1) % 8: slots q%4 -> XCD pair
idx = torch.arange(N, device=dev)
slot = idx % 128; sweep = idx // 128
grp = (slot % 4)                              # which XCD pair serves this slot (pairs {0,1},{2,3},{4,5},{6,7})
# 4 XCD pairs -> 4 angular quarters of 1024 rays
quarters = order.view(4, N // 4)
pos = sweep * 32 + slot // 4
inter = quarters[grp, pos]
res["xcd_pair_quarters"] = run(rays0[inter].contiguous())
print(json.dumps(res))
