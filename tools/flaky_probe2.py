"""Which per-point kernels show call-to-call differences at scale (4000 calls each)?"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from egonerf_amd.synth import build_model
dev = torch.device("cuda", 0)
cfg = synth.SceneConfig(n_voxel=20 ** 3)
model = build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
u = torch.from_numpy(synth.hash_uniform(99, 0, 512 * 7).reshape(512, 7).astype(np.float32))
q = u * 2.6 - 1.3
q[:, 6] = (u[:, 6] > 0.5).float()
q = q.to(dev)
q_in = q.clone(); q_in[:, :6] = q_in[:, :6].clamp(-0.999, 0.999)
q_one = q_in.clone(); q_one[:, 6] = 0
rays = torch.from_numpy(synth.make_rays(256, seed=7)).to(dev)
vd = rays[:, 3:6].repeat(2, 1).contiguous()
feat = model.compute_appfeature(q_in)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
def appf(prec, tab, pts):
    def f():
        model.mlp_precision, model.app_table_dtype = prec, tab
        r = model.compute_appfeature(pts)
        model.mlp_precision, model.app_table_dtype = "f16x3", "f32"
        return r
    return f
cases = [("app f16x3 team (mixed grids)", appf("f16x3", "f32", q_in)), ("app f16x3 team (one grid)", appf("f16x3", "f32", q_one)),
         ("app f32-MFMA (non-team)", appf("f32", "f32", q_in)), ("app f16 tables (non-team)", appf("f16x3", "f16", q_in)),
         ("renderModule", lambda: model.renderModule(q_in, vd, feat)), ("densityfeature", lambda: model.compute_densityfeature(q_in)),
         ("forward 24", lambda: model(rays, n_coarse=24, exp_sampling=True)[0])]
with torch.no_grad():
    for name, fn in cases:
        ref = fn().clone(); bad = 0
        for _ in range(reps):
            bad += int(not torch.equal(fn(), ref))
        print(f"{name:32s} {bad} / {reps}")
