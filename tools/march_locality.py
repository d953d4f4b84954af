"""How much of the density march is cache misses?  The bench batch (4096 random rays x 512) against one ray repeated 4096 times and
64 rays repeated (every tap L1 / L2 resident).  MI355X: 0.1045 / 0.0906 / 0.0919 ms - misses cost 13 % at most."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from egonerf_amd import synth, _lib
dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
N, S = 4096, 512
lib, st, sc = _lib.load(), _lib.stream_handle(), model.scene()
sched = model._sched(S, dev)
z = torch.empty(N, S, device=dev); w = torch.empty_like(z); alpha = torch.empty_like(z); bg = torch.empty(N, device=dev); crd = torch.empty(N, S, 4, device=dev)
def t(rays, reps=30):
    f = lambda: _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), alpha.data_ptr(), 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "m")
    for _ in range(200): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
r = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
same = r[:1].expand(N, 6).contiguous()
# 64 distinct rays repeated: fits L2
few = r[:64].repeat(N // 64, 1).contiguous()
print(json.dumps(dict(random=t(r), one_ray=t(same), rays64=t(few))))
