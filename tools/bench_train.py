"""BASELINE config 4: one training step = forward + backward + Adam on 8192 rays x (128 coarse + 128 fine) samples,
barbershop-size grid, MSE loss against random targets (train.py:245-330 semantics; grids lr 0.02, nets lr 1e-3)."""
import sys, os, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from egonerf_amd.synth import build_model as make_model

dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = make_model(cfg, synth.make_weights(cfg, seed=1234), dev)
model.train()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rays = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
gt = torch.from_numpy(synth.hash_uniform(3, 0, N * 3).reshape(N, 3).astype(np.float32)).to(dev)
from egonerf_amd.optim import FusedAdam
from egonerf_amd.losses import TVLoss, ray_entropy_loss
full = len(sys.argv) > 3 and sys.argv[3] == "full"   # + the Ricoh configs' regularisers (TV density/app, L1, ortho, ray entropy)
opt = (torch.optim.Adam if len(sys.argv) > 3 and sys.argv[3] == "torch_adam" else FusedAdam)(model.get_optparam_groups(0.02, 1e-3),
                                                                                          betas=(0.9, 0.99))
tv = TVLoss()
kw = dict(is_train=True, n_coarse=128, n_fine=128, exp_sampling=True, resampling=True, use_coarse_sample=True)

def step():
    jit = torch.rand(N, 128, device=dev)
    u = torch.rand(N, 128, device=dev)
    rgb, _, _, _, alpha = model(rays, jitter=jit, u=u, **kw)
    loss = torch.mean((rgb - gt) ** 2)
    if full:
        loss = loss + 1e-4 * model.vector_comp_diffs() + 8e-5 * model.density_L1() + 0.1 * model.TV_loss_density(tv) \
            + 0.01 * model.TV_loss_app(tv) + 1e-3 * ray_entropy_loss(alpha)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    model.update_coarse_sigma_grid()  # every step when resampling (train.py:356-357)
    return loss

for _ in range(3):
    step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
t0 = time.perf_counter()
losses = [float(step()) for _ in range(steps)]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# phase split
ev[0].record(); rgb, *_ = model(rays, jitter=torch.rand(N, 128, device=dev), u=torch.rand(N, 128, device=dev), **kw); loss = torch.mean((rgb - gt) ** 2)
ev[1].record(); opt.zero_grad(set_to_none=True); loss.backward()
ev[2].record(); opt.step(); model.update_coarse_sigma_grid()
ev[3].record(); torch.cuda.synchronize()
print(json.dumps(dict(config="train step: %d rays x (128+128), fwd+bwd+Adam%s" % (N, " + TV/L1/ortho/entropy" if full else ""), optimizer=type(opt).__name__, ms_per_step=dt * 1e3, rays_per_s=N / dt,
                      fwd_ms=ev[0].elapsed_time(ev[1]), bwd_ms=ev[1].elapsed_time(ev[2]), adam_ms=ev[2].elapsed_time(ev[3]),
                      loss_first=losses[0], loss_last=losses[-1], peak_mem_GB=torch.cuda.max_memory_allocated() / 2 ** 30)))
