"""Times the sorted-scatter kernels alone on a real training batch's coordinates (config 4 size) - for rocprofv3 --pmc / --kernel-trace."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth, _lib
from egonerf_amd.train import _grad_struct, table_params
dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev); model.train()
N, S = 8192, 256
rays = torch.from_numpy(synth.make_rays(N, seed=1)).to(dev)
lib, st = _lib.load(), _lib.stream_handle()
sc = model.scene(training=True)
# coordinates of a real step: coarse march -> pdf merge -> fine march
sched = model._sched(128, dev)
f = lambda *s: torch.empty(*s, device=dev)
zc, wc, z, w, bg, crd = f(N, 128), f(N, 128), f(N, S), f(N, S), f(N), f(N, S, 4)
near = float(model.near_far[0])
_lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, 128, None, sched.data_ptr(), None, near, 1, zc.data_ptr(), None, 0, wc.data_ptr(), None, None, None, None, st), "m1")
_lib.check(lib.ego_sample_pdf_merge(zc.data_ptr(), wc.data_ptr(), None, N, 128, 128, 1, z.data_ptr(), None, st), "pdf")
_lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, z.data_ptr(), None, None, near, 2, None, None, 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "m2")
dfeat = torch.randn(N, S, device=dev); dv = torch.randn(N * S * 144, device=dev)
dens, app = table_params(model, "density"), table_params(model, "app")
gd, ga = [torch.zeros_like(p) for p in dens], [torch.zeros_like(p) for p in app]
sd, sa = _grad_struct(gd), _grad_struct(ga)
nb = lib.ego_scatter_sorted_workspace_bytes(sc, N, S)
ws = torch.empty(nb, device=dev, dtype=torch.uint8)
def run(which):
    if which == "sort": _lib.check(lib.ego_scatter_sort(sc, crd.data_ptr(), N, S, ws.data_ptr(), nb, st), "sort")
    if which == "dens": _lib.check(lib.ego_scatter_density_sorted(sc, C.byref(sd), crd.data_ptr(), dfeat.data_ptr(), N, S, ws.data_ptr(), nb, st), "d")
    if which == "app": _lib.check(lib.ego_scatter_app_sorted(sc, C.byref(sa), crd.data_ptr(), dv.data_ptr(), None, N, S, ws.data_ptr(), nb, st), "a")
    if which == "dens_atomic": _lib.check(lib.ego_scatter_density(sc, C.byref(sd), crd.data_ptr(), dfeat.data_ptr(), N, S, st), "d")
    if which == "app_atomic": _lib.check(lib.ego_scatter_app(sc, C.byref(sa), crd.data_ptr(), dv.data_ptr(), N, S, st), "a")
def timed(which, label):
    for _ in range(3): run(which)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run(which)
    e1.record(); torch.cuda.synchronize()
    print(f"{label:28s} {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
timed("sort", "sort")
# round 6: the one-pass (fused, fixed-point lines) form against the two-pass form of round 5, and the fused kernel's workgroup shapes
ref = None
for label, env in (("separate", dict(EGO_SORTED_LINES="separate")), ("fused nw16", dict(EGO_FUSED_NW="16")), ("fused nw12", dict(EGO_FUSED_NW="12")),
                   ("fused nw8", dict(EGO_FUSED_NW="8")), ("separate again", dict(EGO_SORTED_LINES="separate")), ("fused default", {})):
    for k in ("EGO_SORTED_LINES", "EGO_FUSED_NW"): os.environ.pop(k, None)
    os.environ.update(env)
    for which in ("dens", "app"):
        timed(which, f"{which} {label}")
    torch.cuda.synchronize()
    cur = [g.clone() for g in gd + ga]
    if ref is None:
        ref = cur
    else:
        worst = max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30) for a, b in zip(cur, ref))
        print(f"    worst |diff| vs the separate form, relative to each table's max: {worst:.2e}", flush=True)
for k in ("EGO_SORTED_LINES", "EGO_FUSED_NW"): os.environ.pop(k, None)
for which in ("dens_atomic", "app_atomic"):
    timed(which, which)
# cell-size distribution of the three sorts (torch restatement of cell_of / k_sort_keys)
if os.environ.get("PROBE_CELLS"):
    res = [int(v) for v in model.gridSize.tolist()]
    c = crd.view(-1, 4)
    g = (c[:, 3] != 0).long()
    def cell(x, n):
        i0 = torch.floor((x + 1.0) * (0.5 * (n - 1))).clamp(-2, n).long()
        return torch.where((i0 < -1) | (i0 > n - 1), torch.full_like(i0, -1), i0 + 1)
    cr, cth, cph = cell(c[:, 0], res[0]), cell(c[:, 1], res[1]), cell(c[:, 2], res[2])
    for name, (a, na), (b, nb) in (("sort0 (phi,r) plane1", (cph, res[2]), (cr, res[0])), ("sort1 (r,theta) plane0", (cr, res[0]), (cth, res[1])), ("sort2 (theta,phi) plane2", (cth, res[1]), (cph, res[2]))):
        ok = (a >= 0) & (b >= 0)
        key = ((g * (na + 1) + a) * (nb + 1) + b)[ok]
        cnt = torch.bincount(key, minlength=2 * (na + 1) * (nb + 1))
        nz = cnt[cnt > 0].float()
        q = torch.quantile(nz, torch.tensor([0.5, 0.9, 0.99, 0.999], device=nz.device)).tolist()
        top = torch.sort(cnt, descending=True).values[:8].tolist()
        print(f"{name}: cells {cnt.numel()}, non-empty {nz.numel()}, samples {int(nz.sum())}, mean {nz.mean():.1f}, median/p90/p99/p99.9 {q}, largest {top}, samples in cells > 256: {int(cnt[cnt > 256].sum())}, > 1024: {int(cnt[cnt > 1024].sum())}")
