"""A/B of the matrix-product arithmetics on the bench workload (4096 rays x 512 samples): kernel time of ego_shade and the
composited / per-sample error of each mode against the oracle in float64.   python tools/prec_compare.py [n_err_rays]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth, _lib
from oracle.egonerf_oracle import OracleScene

dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
w = synth.make_weights(cfg, seed=1234)
model = synth.build_model(cfg, w, dev)
N, S = 4096, 512
rays = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
n_err = int(sys.argv[1]) if len(sys.argv) > 1 else 256
o64 = OracleScene(cfg, w, dtype=torch.float64)
with torch.no_grad():
    ref, inter = o64.forward(rays[:n_err].cpu(), n_coarse=S, keep=True)
lib, st = _lib.load(), _lib.stream_handle()
res = {}
for prec in ("f16x3", "f16f8", "f32", "f16x3", "f16f8"):
    model.mlp_precision = prec
    sc = model.scene()
    sched = model._sched(S, dev)
    z = torch.empty(N, S, device=dev); alpha = torch.empty_like(z); wgt = torch.empty_like(z); bg = torch.empty(N, device=dev)
    crd = torch.empty(N, S, 4, device=dev); rgb = torch.empty(N, S, 3, device=dev)
    _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), alpha.data_ptr(), 0,
                                     wgt.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march")
    reps = 50
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(3):
        _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N, S, rgb.data_ptr(), None, None, st), "shade")
    ev[0].record()
    for i in range(reps):
        _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N, S, rgb.data_ptr(), None, None, st), "shade")
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))
    with torch.no_grad():
        out = model(rays[:n_err], n_coarse=S, exp_sampling=True)
    e_comp = float((out[0].cpu().double() - ref[0]).abs().max())
    e_samp = float((rgb[:n_err].cpu().double() - inter["rgb_samples"]).abs().max())
    res.setdefault(prec, []).append(dict(shade_ms=ms, max_rgb_err_composited=e_comp, max_rgb_err_per_sample=e_samp, finite=bool(torch.isfinite(rgb).all())))
    print(prec, res[prec][-1], flush=True)
print(json.dumps(res))
