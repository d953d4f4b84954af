"""Per-kernel device time of one backward pass at config 4 (8192 rays x (128+128)), via torch.profiler."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from egonerf_amd import synth
from egonerf_amd.synth import build_model as make_model
dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = make_model(cfg, synth.make_weights(cfg, seed=1234), dev); model.train()
N = 8192
rays = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
gt = torch.from_numpy(synth.hash_uniform(3, 0, N * 3).reshape(N, 3).astype(np.float32)).to(dev)
kw = dict(is_train=True, n_coarse=128, n_fine=128, exp_sampling=True, resampling=True, use_coarse_sample=True)
jit = torch.rand(N, 128, device=dev); u = torch.rand(N, 128, device=dev)
from torch.profiler import profile, ProfilerActivity
for it in range(3):
    if it == 2:
        with profile(activities=[ProfilerActivity.CUDA]) as fprof:
            rgb, *_ = model(rays, jitter=jit, u=u, **kw); torch.cuda.synchronize()
    else:
        rgb, *_ = model(rays, jitter=jit, u=u, **kw)
    loss = torch.mean((rgb - gt) ** 2)
    model.zero_grad(set_to_none=True)
    if it == 2:
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            loss.backward(); torch.cuda.synchronize()
    else:
        loss.backward()
print("---- forward")
for r in sorted(fprof.key_averages(), key=lambda e: -e.device_time_total)[:8]:
    print(f"{r.device_time_total / 1000:8.3f} ms  x{r.count:<3d} {r.key[:110]}")
print("total device ms:", round(sum(r.device_time_total for r in fprof.key_averages()) / 1000, 2))
print("---- backward")
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:20]
for r in rows:
    print(f"{r.device_time_total / 1000:8.3f} ms  x{r.count:<3d} {r.key[:110]}")
print("total device ms:", round(sum(r.device_time_total for r in prof.key_averages()) / 1000, 2))
