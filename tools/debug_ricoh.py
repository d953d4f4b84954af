"""Localises a HIP-vs-reference mismatch on the Ricoh scene (config 3): per-ray errors, then per-stage comparison vs the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from oracle.egonerf_oracle import OracleScene
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ricoh.npz"))
cfg = synth.SceneConfig(**synth.RICOH)
w = synth.make_weights(cfg, seed=1234)
model = synth.build_model(cfg, w, "cuda")
orc = OracleScene(cfg, w)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rays = torch.from_numpy(fx[f"rays/{k}"])
for prec in ("f16x3", "f32"):
    model.mlp_precision = prec
    with torch.no_grad():
        o = model(rays.cuda(), n_coarse=S, exp_sampling=True)
    err = (o[0].cpu().numpy() - fx[f"nr512/{k}/rgb"]) if S == 512 else None
    if err is not None:
        e = np.abs(err).max(1)
        bad = np.where(e > 1e-4)[0]
        print(prec, "max rgb err", e.max(), "bad rays", len(bad), bad[:20], "depth err", np.abs(o[1].cpu().numpy() - fx[f"nr512/{k}/depth"]).max())
with torch.no_grad():
    ref, inter = orc.forward(rays, n_coarse=S, keep=True)
print("oracle vs golden", float((ref[0] - torch.from_numpy(fx[f"nr512/{k}/rgb"])).abs().max()) if S == 512 else "")
model.mlp_precision = "f16x3"
r = rays.cuda()
with torch.no_grad():
    xyz, z, _ = model.sample_ray_exp(r[:, :3], r[:, 3:6], is_train=False, N_samples=S)
    c7 = model.coordinates.from_cartesian(xyz)
    c7n = model.coordinates.normalize_coord(c7, downsample=2)
    print("z", float((z.cpu() - inter["z"]).abs().max()), "c7n", float((c7n.cpu() - inter["c7n"]).abs().max()),
          "flag mismatch", int((c7n[..., 6].cpu() != inter["c7n"][..., 6]).sum()))
    c7n_ref = inter["c7n"].cuda()
    sf = model.compute_densityfeature(c7n_ref)
    print("sigma_feat", float((sf.cpu() - inter["sigma_feat"]).abs().max()))
    af = model.compute_appfeature(c7n_ref)
    d = (af.cpu() - inter["app_feat"]).abs()
    print("app_feat", float(d.max()), "at", np.unravel_index(int(d.argmax()), d.shape), "scale", float(inter["app_feat"].abs().max()))
    vd = r[:, 3:6].view(-1, 1, 3).expand(xyz.shape)
    rgb_s = model.renderModule(c7n_ref, vd, inter["app_feat"].cuda())
    d = (rgb_s.cpu() - inter["rgb_samples"]).abs()
    print("mlp on ref feat", float(d.max()))
    rgb_s2 = model.renderModule(c7n_ref, vd, af)
    d2 = (rgb_s2.cpu() - inter["rgb_samples"]).abs()
    print("mlp on hip feat", float(d2.max()), "weighted", float((d2.max(-1)[0] * inter["weight"]).sum(-1).max()))
    worst = int((d2.max(-1)[0] * inter["weight"]).sum(-1).argmax())
    print("worst ray", worst, "feat range there", float(inter["app_feat"][worst].abs().max()), "pre-act |x| max overall", float(inter["app_feat"].abs().max()))
    sw = (d2.max(-1)[0] * inter["weight"])[worst]
    js = torch.topk(sw, 5).indices
    for j in js.tolist():
        print("  sample", j, "w", float(inter["weight"][worst, j]), "rgb err", d2[worst, j].tolist(), "feat max", float(inter["app_feat"][worst, j].abs().max()),
              "c7n", inter["c7n"][worst, j].tolist())
