"""Event-timed ego_march_density in the three shapes the benches use (4096 x 512 headline; 16384 x 128 coarse on the pooled tables and
16384 x 256 fine with explicit distances: the ERP chunk), plus a checksum of the outputs so that variants can be compared."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egonerf_amd import _lib, synth
lib, st = _lib.load(), _lib.stream_handle()
dev = "cuda"
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
sc = model.scene()
def run(N, S, coarse, z_in=None, reps=100):
    rays = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
    sched = model._sched(S, dev)
    z, w, bg, crd = torch.empty(N, S, device=dev), torch.empty(N, S, device=dev), torch.empty(N, device=dev), torch.empty(N, S, 4, device=dev)
    act = torch.empty(N * S // 32 + 1, device=dev, dtype=torch.uint8)
    def go():
        _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None if z_in is None else z_in.data_ptr(), None if z_in is not None else sched.data_ptr(), None,
                                         cfg.near, coarse, None if z_in is not None else z.data_ptr(), None, 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, act.data_ptr(), st), "march")
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        for _ in range(8): go()
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        ev[i].record(); go()
    ev[reps].record(); torch.cuda.synchronize()
    ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)])
    return float(np.median(ms)), float(w.double().sum()), float(crd.double().abs().sum()), (z if z_in is None else z_in)
a = run(4096, 512, 0)
b = run(16384, 128, 1)
zf = torch.sort(torch.cat([b[3], b[3] + 0.004], 1), 1).values.contiguous()
c = run(16384, 256, 2, z_in=zf)
print(f"march 4096x512 {a[0]*1e3:.1f} us | 16384x128 coarse {b[0]*1e3:.1f} us | 16384x256 fine {c[0]*1e3:.1f} us | checksums {a[1]:.6f} {a[2]:.3f} {b[1]:.6f} {c[1]:.6f}")
if len(sys.argv) > 1 and sys.argv[1] == "sweep":   # wave-slot quantisation: time against the number of rays (one wave per ray, 3 waves per SIMD = 3072 slots)
    for n in (1024, 2048, 3072, 3584, 4096, 5120, 6144, 8192, 9216, 12288):
        print(f"  N = {n:6d} x 512: {run(n, 512, 0, reps=50)[0] * 1e3:7.1f} us   ({n / 3072:.2f} rounds of wave slots)")
