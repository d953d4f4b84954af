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
    global ws, nb
    if which == "sort": _lib.check(lib.ego_scatter_sort(sc, crd.data_ptr(), N, S, ws.data_ptr(), nb, st), "sort")
    if which == "dens": _lib.check(lib.ego_scatter_density_sorted(sc, C.byref(sd), crd.data_ptr(), dfeat.data_ptr(), N, S, ws.data_ptr(), nb, st), "d")
    if which == "app": _lib.check(lib.ego_scatter_app_sorted(sc, C.byref(sa), crd.data_ptr(), dv.data_ptr(), None, None, None, 0, N, S, ws.data_ptr(), nb, st), "a")
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
if os.environ.get("PROBE_PROF"):   # a -DEGO_WALK_PROF build (tools/build_variant.sh): per-phase cycles of the walk kernel
    import ctypes
    raw = ctypes.CDLL(_lib.LIB)
    buf = (ctypes.c_ulonglong * 20)()
    os.environ["EGO_SORTED_LINES"] = os.environ.get("EGO_SORTED_LINES", "separate")
    for which in ("dens", "app"):
        run(which); torch.cuda.synchronize(); raw.ego_debug_walk_prof(buf)
        run(which); torch.cuda.synchronize(); raw.ego_debug_walk_prof(buf)
        v = list(buf)[0:8] if which == "dens" else list(buf)[8:16]
        steps, iters = max(v[5], 1), max(v[6], 1)
        names = ["top+stage1 (per step)", "issue (per iter)", "wait (per iter)", "stage 2 total (per step)", "tail (per step)", "steps", "iterations", "chunk prologue (total)"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(which); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        raw.ego_debug_walk_prof(buf)
        busy = buf[18]
        print(which, "wave-steps", v[5], "iterations", v[6], f"| call {ms:.3f} ms; slowest wave {busy} ticks, mean wave {sum(v[i] for i in (0, 3, 4, 7)) / (256 * (16 if which == 'dens' else 12)):.0f} ticks")
        wv = (ctypes.c_uint32 * (4096 * 4))()
        raw.ego_debug_walk_waves(wv)
        W_ = np.array(list(wv), dtype=np.int64).reshape(4096, 4)
        W_ = W_[W_[:, 1] > 0]
        for pp in range(6):
            x = W_[W_[:, 3] == pp]
            if len(x):
                print(f"   pair {pp}: waves {len(x)}, steps/wave mean {x[:, 1].mean():.1f} max {x[:, 1].max()}, iterations/wave mean {x[:, 2].mean():.1f} max {x[:, 2].max()}, "
                      f"busy mean {x[:, 0].mean():.0f} max {x[:, 0].max()} ticks; ticks/iteration mean {(x[:, 0] / x[:, 2]).mean():.0f} max {(x[:, 0] / x[:, 2]).max():.0f}")
        print("   clk per step: top+stage1 %.0f, stage2 %.0f, tail %.0f | per iteration: issue %.0f, wait %.0f, rest %.0f | chunk prologues total %.0f clk per step" % (
            v[0] / steps, v[3] / steps, v[4] / steps, v[1] / iters, v[2] / iters, (v[3] - v[1] - v[2]) / iters, v[7] / steps))
    sys.exit(0)
if os.environ.get("PROBE_PAIRS"):   # the walk with ONE (sort, grid) pair active at a time (same deal of workgroups): where does the time go?
    for dbg in [0] + [16 * (p + 1) for p in range(6)] + [0]:
        os.environ["EGO_FUSED_DBG"] = str(dbg)
        timed("dens", f"dens pair={dbg // 16 - 1}"); timed("app", f"app pair={dbg // 16 - 1}")
    sys.exit(0)
if os.environ.get("PROBE_ONLY"):   # tools/sorted_kernels.sh: the current environment's form only, for a kernel trace
    timed("dens", "dens"); timed("app", "app")
    sys.exit(0)
# round 6: the one-pass (fused, fixed-point lines) form against the two-pass form of round 5, and the fused kernel's workgroup shapes
ref = None
ENV_KEYS = ("EGO_SORTED_LINES", "EGO_SORTED_WALK", "EGO_FUSED_DBG")
for label, env in (("r05 form", dict(EGO_SORTED_WALK="0")), ("walk", {}), ("walk, lines separate", dict(EGO_SORTED_LINES="separate")),
                   ("r05 form again", dict(EGO_SORTED_WALK="0")),
                   ("walk again", {})):
    for k in ENV_KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    nb = lib.ego_scatter_sorted_workspace_bytes(sc, N, S)          # the key layout (line blocks) follows the environment: size and sort again
    ws = torch.empty(nb, device=dev, dtype=torch.uint8)
    run("sort")
    for which in ("dens", "app"):
        timed(which, f"{which} {label}")
    torch.cuda.synchronize()
    cur = [g.clone() for g in gd + ga]
    if ref is None:
        ref = cur
    else:
        worst = max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30) for a, b in zip(cur, ref))
        print(f"    worst |diff| vs the r05 form, relative to each table's max: {worst:.2e}", flush=True)
for k in ENV_KEYS: os.environ.pop(k, None)
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
        steps = int(((cnt + 15) // 16).sum())
        # lockstep quads of 4 consecutive cells (r05): batches = ceil(max of 4 / 16); independent groups (walk): per 64-cell chunk, max over the 4 groups of their 16 cells' steps
        c4 = cnt[: cnt.numel() // 4 * 4].view(-1, 4)
        quad_batches = int(((c4.max(dim=1).values + 15) // 16).sum())
        c64 = ((cnt[: cnt.numel() // 64 * 64] + 15) // 16).view(-1, 4, 16).sum(dim=2)
        walk_steps = int(c64.max(dim=1).values.sum())
        print(f"    group-steps {steps} (/4 = {steps / 4:.0f}), r05 quad-batches {quad_batches}, walk wave-steps {walk_steps}")
        print(f"{name}: cells {cnt.numel()}, non-empty {nz.numel()}, samples {int(nz.sum())}, mean {nz.mean():.1f}, median/p90/p99/p99.9 {q}, largest {top}, samples in cells > 256: {int(cnt[cnt > 256].sum())}, > 1024: {int(cnt[cnt > 1024].sum())}")
