"""Does capturing the whole training step (forward + backward + FusedAdam + coarse-table refresh) in one hipGraph pay?  Times the
bench's 8192-ray step eagerly and as torch.cuda.CUDAGraph replays on the same box, and the host time of an eager step (how far
the Python / ctypes side runs ahead of the device).  Timing probe: the captured optimiser step keeps the lr / step count of the
capture (FusedAdam takes them as launch arguments)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egonerf_amd import synth
from egonerf_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
N, NC, NF = 8192, 128, 128
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
model.train()
rays = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
gt = torch.from_numpy(synth.hash_uniform(3, 0, N * 3).reshape(N, 3).astype(np.float32)).to(dev)
opt = FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
kw = dict(is_train=True, n_coarse=NC, n_fine=NF, exp_sampling=True, resampling=True, use_coarse_sample=True)
def step():
    rgb, _, _, _, alpha = model(rays, jitter=torch.rand(N, NC, device=dev), u=torch.rand(N, NF, device=dev), **kw)
    loss = torch.mean((rgb - gt) ** 2)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    model.update_coarse_sigma_grid()
    return loss
def timeit(fn, k=20, ramp=0.3):
    t_end = time.perf_counter() + ramp
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(k): fn()
    t_host = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k, t_host / k * 1e3
from egonerf_amd.train import GraphedTrainStep
out = {"eager_ms": [], "graph_ms": [], "eager_host_ms": [], "graph_host_ms": []}
a, b = timeit(step)
out["eager_ms"].append(round(a, 4)); out["eager_host_ms"].append(round(b, 4))
opt2 = FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), capturable=True, lr_factor=0.1 ** (1 / 30000))
graphed = GraphedTrainStep(model, opt2, rays, gt, {k: v for k, v in kw.items() if k != "is_train"}, warmup=2)
for rep in range(3):   # alternate on one box: graph replay / eager
    a, b = timeit(lambda: graphed(rays, gt))
    out["graph_ms"].append(round(a, 4)); out["graph_host_ms"].append(round(b, 4))
    a, b = timeit(step)
    out["eager_ms"].append(round(a, 4)); out["eager_host_ms"].append(round(b, 4))
out["loss"] = float(graphed.loss)
print(json.dumps(out))
