import numpy as np, torch, sys, ctypes as C
sys.path.insert(0,'/root/repo')
from egonerf_amd import synth, _lib
from egonerf_amd import train as T
from egonerf_amd.synth import build_model as make_model
cfg=synth.SceneConfig()
w=synth.make_weights(cfg, seed=1234)
model=make_model(cfg, w, 'cuda')
rays=torch.from_numpy(synth.make_rays(128, seed=3)).cuda()
jit=torch.from_numpy(synth.hash_uniform(8,0,128*32).reshape(128,32).astype(np.float32)).cuda()
u=torch.from_numpy(synth.hash_uniform(8,1,128*32).reshape(128,32).astype(np.float32)).cuda()
gt=torch.from_numpy(synth.hash_uniform(8,2,128*3).reshape(128,3).astype(np.float32)).cuda()
saved={}
orig=T.RenderFunction.backward
def hook(ctx, g, *a):
    saved.update({k:(v.clone() if torch.is_tensor(v) else v) for k,v in ctx.saved.items()})
    saved['g']=g.clone()
    return orig(ctx, g, *a)
T.RenderFunction.backward=staticmethod(hook)
for resamp in (False, True):
    model.zero_grad()
    kw=dict(n_coarse=32, n_fine=32, resampling=True, u=u) if resamp else dict(n_coarse=64)
    rgb,*_=model(rays, is_train=True, exp_sampling=True, jitter=(jit if resamp else torch.cat([jit,jit],1)), **kw)
    torch.mean((rgb-gt)**2).backward()
    # re-run the backward kernels by hand to get the buffers
    lib,st=_lib.load(),_lib.stream_handle(); sc=model.scene(); N,S=128,64; M=N*S
    f=lambda *s: torch.empty(*s, device='cuda')
    g_d=[torch.zeros_like(p) for p in T.table_params(model,'density')]; g_a=[torch.zeros_like(p) for p in T.table_params(model,'app')]
    dc=f(N,S,3); gd=T._grad_struct(g_d)
    _lib.check(lib.ego_march_backward(sc, C.byref(gd), saved['coords'].data_ptr(), saved['z'].data_ptr(), saved['alpha'].data_ptr(), saved['weight'].data_ptr(), saved['sigma'].data_ptr(), saved['bg'].data_ptr(), saved['rgb'].data_ptr(), saved['g'].contiguous().data_ptr(), saved['raw'].data_ptr(), None, N,S, dc.data_ptr(), st),'mb')
    dc_in=dc.clone()
    tp=f(lib.ego_train_packed_floats()); _lib.check(lib.ego_pack_train(sc,tp.data_ptr(),st),'pt')
    dh2,dh1,dfe=f(M,128),f(M,128),f(M,64); ga=T._grad_struct(g_a)
    ds=_lib.ShadeDump(saved['x'].data_ptr(),saved['h1'].data_ptr(),saved['h2'].data_ptr(),saved['v'].data_ptr())
    _lib.check(lib.ego_shade_backward(sc,tp.data_ptr(),C.byref(ga),saved['coords'].data_ptr(),dc.data_ptr(),saved['rgb'].data_ptr(),C.byref(ds),dh2.data_ptr(),dh1.data_ptr(),dfe.data_ptr(),N,S,st),'sb')
    torch.cuda.synchronize()
    hid=T._layout(1,128,'cuda'); mlp=model.renderModule.mlp
    c=saved['rgb'].view(M,3); do_exp=(dc_in.view(M,3)*c*(1-c)).double()
    print('resampling',resamp,'do err', float((dc.view(M,3).double()-do_exp).abs().max()/do_exp.abs().max()))
    W3=mlp[4].weight.detach().double(); W2=mlp[2].weight.detach().double()
    dh2_exp=(saved['h2']>0).double()*(do_exp@W3[:,hid])
    print('  dh2 err', float((dh2.double()-dh2_exp).abs().max()/dh2_exp.abs().max()), 'scale', float(dh2_exp.abs().max()))
    W2l=W2[hid][:,hid]  # [v_lane][u_lane]
    dh1_exp=(saved['h1']>0).double()*(dh2_exp@W2l)
    e=(dh1.double()-dh1_exp).abs()
    print('  dh1 err', float(e.max()/dh1_exp.abs().max()), 'worst row', int(e.max(1)[0].argmax()), 'row rel err', float((e.max(1)[0]/dh1_exp.abs().max(1)[0].clamp_min(1e-30)).max()))
    rowrel=(e.max(1)[0]/dh1_exp.abs().max(1)[0].clamp_min(1e-30))
    bad=(rowrel>1e-3).nonzero().flatten()
    print('  rows with rel err>1e-3:', len(bad), bad[:10].tolist(), 'their max|dh2|', dh2_exp.abs().max(1)[0][bad[:5]].tolist())
