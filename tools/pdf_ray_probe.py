"""Where does ONE ray's resampling differ between the HIP path and the oracles?  Coarse weights (HIP march on the pooled tables vs
oracle), the cdf tail, and the fine samples of ego_sample_pdf_merge vs OracleScene.sample_pdf in float32 / float64.
  python tools/pdf_ray_probe.py <seed> <case> <ray>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egonerf_amd import _lib
from tests.helpers import campaign_cases, make_model, make_oracle
seed, want, ray = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.set_printoptions(precision=9, linewidth=200)
for case, cfg, w, rays, kw in campaign_cases(seed, want + 1):
    if case != want:
        continue
    o32, o64 = make_oracle(cfg, w), make_oracle(cfg, w, dtype=torch.float64)
    with torch.no_grad():
        _, i32 = o32.forward(rays, keep=True, **kw)
        _, i64 = o64.forward(rays.double(), keep=True, **kw)
    model = make_model(cfg, w, "cuda")
    lib, st, sc, dev = _lib.load(), _lib.stream_handle(), model.scene(), "cuda"
    N, Sc, nf = rays.shape[0], kw["n_coarse"], kw["n_fine"]
    r = rays.cuda().contiguous()
    sched = model._sched(Sc, dev)
    zc, wc = torch.empty(N, Sc, device=dev), torch.empty(N, Sc, device=dev)
    _lib.check(lib.ego_march_density(sc, r.data_ptr(), N, Sc, None, sched.data_ptr(), None, float(model.near_far[0]), 1, zc.data_ptr(), None, 0,
                                     wc.data_ptr(), None, None, None, None, st), "march")
    zn = torch.empty(N, nf, device=dev); zo = torch.empty(N, nf, device=dev)
    _lib.check(lib.ego_sample_pdf_merge(zc.data_ptr(), wc.data_ptr(), None, N, Sc, nf, 0, zo.data_ptr(), zn.data_ptr(), st), "pdf")
    b = ray
    cw32, cw64, cwh = i32["coarse_weight"][b], i64["coarse_weight"][b], wc[b].cpu()
    print("coarse weights: max |HIP - f32 oracle|", float((cwh - cw32).abs().max()), " max |f32 - f64|", float((cw32.double() - cw64).abs().max()), " sum", float(cw64.sum()))
    for name, cw in (("f32 oracle", cw32), ("f64 oracle", cw64), ("HIP weights in the f32 formula", cwh)):
        ww = cw[1:-1] + 1e-5
        pdf = ww / ww.sum()
        cdf = torch.cumsum(pdf, 0)
        print(f"  {name:32s} total {float(ww.sum()):.9g}  pdf[-3:] {pdf[-3:].tolist()}  cdf[-3:] {[float(x) for x in cdf[-3:]]}  cdf[-1] > 1: {bool(cdf[-1] > 1)}")
    zh, z32, z64 = zn[b].cpu(), i32["z_new"][b], i64["z_new"][b]
    if nf <= 4:
        print("z_new  f32 oracle", z32.tolist(), " f64 oracle", z64.tolist(), " HIP", zh.tolist())
    else:
        d_h, d_32 = (zh.double() - z64).abs(), (z32.double() - z64).abs()
        print("fine samples farther than 1e-5 from the f64 oracle: HIP", [(i, round(float(d_h[i]), 6)) for i in torch.nonzero(d_h > 1e-5).flatten().tolist()],
              " f32 oracle", [(i, round(float(d_32[i]), 6)) for i in torch.nonzero(d_32 > 1e-5).flatten().tolist()])
    zmid = 0.5 * (zc[b, 1:] + zc[b, :-1]).cpu()
    print("last three bin midpoints", zmid[-3:].tolist())
