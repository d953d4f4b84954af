import faulthandler, sys, os, time
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egonerf_amd import synth, train
from egonerf_amd.optim import FusedAdam
from egonerf_amd.train import GraphedTrainStep
print("SIDE_STREAM_SCATTER", train.SIDE_STREAM_SCATTER, flush=True)
dev = "cuda"
n_vox = float(sys.argv[1]) if len(sys.argv) > 1 else 20 ** 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = synth.SceneConfig(n_voxel=n_vox)
m = synth.build_model(cfg, synth.make_weights(cfg, seed=9), dev); m.train()
o = FusedAdam(m.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), capturable=True)
rays = torch.from_numpy(synth.make_rays(N, seed=2)).to(dev); gt = torch.rand(N, 3, device=dev)
NC = int(os.environ.get("PROBE_NC", "32"))
kw = dict(n_coarse=NC, n_fine=NC, exp_sampling=True, resampling=True, use_coarse_sample=True)
# eager first
if os.environ.get("PROBE_EAGER_FIRST"):
    rgb, *_ = m(rays, is_train=True, **kw); torch.mean((rgb - gt) ** 2).backward(); torch.cuda.synchronize(); print("eager ok", flush=True)
g = GraphedTrainStep(m, o, rays, gt, kw, warmup=2)
torch.cuda.synchronize(); print("captured", flush=True)
for i in range(3):
    l = g(rays, gt); torch.cuda.synchronize(); print("replay", i, float(l), flush=True)
if os.environ.get("PROBE_EAGER_AFTER"):
    for i in range(3):
        rgb, *_ = m(rays, is_train=True, jitter=torch.rand(N, kw["n_coarse"], device=dev), u=torch.rand(N, kw["n_fine"], device=dev), **kw)
        loss = torch.mean((rgb - gt) ** 2)
        o.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize(); print("eager-after bwd", i, float(loss), flush=True)
        o.step(); m.update_coarse_sigma_grid()
        torch.cuda.synchronize(); print("eager-after step", i, flush=True)

if os.environ.get("PROBE_ASYNC"):
    def step():
        rgb, *_ = m(rays, is_train=True, jitter=torch.rand(N, kw["n_coarse"], device=dev), u=torch.rand(N, kw["n_fine"], device=dev), **kw)
        loss = torch.mean((rgb - gt) ** 2)
        o.zero_grad(set_to_none=True)
        loss.backward()
        o.step(); m.update_coarse_sigma_grid()
    mode = os.environ["PROBE_ASYNC"]
    g(rays, gt)
    if mode == "sync_after_replay": torch.cuda.synchronize()
    for i in range(8):
        step()
        if mode == "sync_each": torch.cuda.synchronize(); print("async step", i, flush=True)
    torch.cuda.synchronize(); print("async mode", mode, "done", flush=True)
