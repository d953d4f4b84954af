import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from egonerf_amd import synth
from tests.helpers import make_model, make_oracle
cfg=synth.SceneConfig()
w=synth.make_weights(cfg, seed=1234)
model=make_model(cfg, w, 'cuda')
rays=torch.from_numpy(synth.make_rays(128, seed=3))
jit=torch.from_numpy(synth.hash_uniform(8,0,128*32).reshape(128,32).astype(np.float32))
u=torch.from_numpy(synth.hash_uniform(8,1,128*32).reshape(128,32).astype(np.float32))
gt=torch.from_numpy(synth.hash_uniform(8,2,128*3).reshape(128,3).astype(np.float32))
for resamp in (False, True):
    model.zero_grad()
    kw=dict(n_coarse=32, n_fine=32, resampling=True, u=u.cuda()) if resamp else dict(n_coarse=64)
    rgb,*_=model(rays.cuda(), is_train=True, exp_sampling=True, jitter=(jit if resamp else torch.cat([jit,jit],1)).cuda(), **kw)
    torch.mean((rgb-gt.cuda())**2).backward()
    oracle=make_oracle(cfg,w)
    for v in oracle.w.values(): v.requires_grad_(True)
    oracle.update_coarse_sigma_grid()
    okw=dict(n_coarse=32, n_fine=32, resampling=True, u=u) if resamp else dict(n_coarse=64)
    ref,*_=oracle.forward(rays, is_train=True, jitter=(jit if resamp else torch.cat([jit,jit],1)), **okw)
    torch.mean((ref-gt)**2).backward()
    d=(rgb.detach().cpu()-ref.detach()).abs()
    print('resampling',resamp,'fwd max diff',float(d.max()),'rays with diff>1e-5:', int((d.max(1)[0]>1e-5).sum()))
    for k,p in model.named_parameters():
        r=oracle.w[k].grad; r=torch.zeros_like(oracle.w[k]) if r is None else r
        g=p.grad.detach().cpu(); sc=float(r.abs().max())
        e=float((g-r).abs().max())/max(sc,1e-12)
        if e>1e-4 or 'mlp' in k or 'basis' in k: print(f"  {k:30s} rel {e:.2e} refmax {sc:.2e}")
