"""Stand-alone grid-sample probes (SURVEY 7.1 last bullet): ego_app_feature and ego_density_feature (fine and
pooled tables) on the bench batch's own sample coordinates (4096 rays x 512 samples = 2 097 152 points), event-timed, with their
algorithmic tap bytes per second.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel rows of profiles/r04/.
    python tools/stage_probe.py [n_voxel]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth, _lib

dev = torch.device("cuda", 0)
cfg = synth.SceneConfig() if len(sys.argv) < 2 else synth.SceneConfig(n_voxel=float(sys.argv[1]))
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
N, S = 4096, 512
M = N * S
lib, st, sc = _lib.load(), _lib.stream_handle(), model.scene()
sched = model._sched(S, dev)
res = dict(grid=cfg.grid, points=M)
batches = []
for b in range(8):   # eight ray batches: the probes never re-read the previous call's texels
    rays = torch.from_numpy(synth.make_rays(N, seed=1 + b)).to(dev)
    z, w, bg, crd = torch.empty(N, S, device=dev), torch.empty(N, S, device=dev), torch.empty(N, device=dev), torch.empty(N, S, 4, device=dev)
    _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), None, 0,
                                     w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march")
    flat = crd.view(M, 4)
    yang = flat[:, 3] != 0
    c7 = torch.zeros(M, 7, device=dev)
    c7[~yang, 0:3] = flat[~yang, 0:3]; c7[yang, 3:6] = flat[yang, 0:3]; c7[:, 6] = flat[:, 3]
    batches.append(c7)
feat = torch.empty(M, 27, device=dev)
dens = torch.empty(M, device=dev)


def timeit(fn, reps=16):
    for i in range(2):
        fn(batches[i % 8])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(batches[i % 8]); ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


t = timeit(lambda c: _lib.check(lib.ego_app_feature(sc, c.data_ptr(), M, feat.data_ptr(), st), "app"))
res["ego_app_feature"] = dict(ms=t, algorithmic_GBps=3456 * M / t / 1e6, frac_of_hbm_peak=3456 * M / t / 1e6 / 8000)
t = timeit(lambda c: _lib.check(lib.ego_density_feature(sc, c.data_ptr(), M, 0, dens.data_ptr(), st), "density"))
res["ego_density_feature"] = dict(ms=t, algorithmic_GBps=1152 * M / t / 1e6, frac_of_hbm_peak=1152 * M / t / 1e6 / 8000)
t = timeit(lambda c: _lib.check(lib.ego_density_feature(sc, c.data_ptr(), M, 1, dens.data_ptr(), st), "density coarse"))
res["ego_density_feature(coarse)"] = dict(ms=t, algorithmic_GBps=1152 * M / t / 1e6, frac_of_hbm_peak=1152 * M / t / 1e6 / 8000)
print(json.dumps(res))
