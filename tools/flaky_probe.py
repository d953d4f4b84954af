"""Stress compute_appfeature / forward for call-to-call differences and print WHERE they are (tile, position in tile, column)."""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
with torch.no_grad():
    for name, pts in (("in-range", q_in), ("scattered", q)):
        ref = model.compute_appfeature(pts).clone()
        n = 0
        slots = np.zeros(8, int); jhist = np.zeros(32, int)
        for it in range(reps):
            o = model.compute_appfeature(pts)
            if not torch.equal(o, ref):
                n += 1
                d = (o != ref)
                rows = d.any(1).nonzero().flatten().cpu().numpy()
                for t_ in np.unique(rows // 32): slots[t_ % 8] += 1
                jhist += np.bincount(rows % 32, minlength=32)
                cols = d.any(0).nonzero().flatten().cpu().numpy()
                if n <= 0:
                    bad = o[rows]
                    dist = torch.cdist(bad.double(), ref.double())          # [bad rows][all ref rows]
                    near = dist.argmin(1).cpu().numpy()
                    print("   bad rows", rows.tolist(), "nearest ref rows", near.tolist(), "at distance", [round(float(x), 5) for x in dist.min(1).values],
                          "distance to own ref", [round(float(x), 5) for x in (bad - ref[rows]).norm(dim=1)])
                    print("   ratio bad/ref of row", int(rows[0]), ":", [round(float(x), 3) for x in (bad[0] / ref[rows[0]])[:12]])
                if n <= 0:
                    print(name, "call", it, "rows", len(rows), "tiles", np.unique(rows // 32), "j", rows % 32, "feature cols", cols,
                          "max diff", float((o - ref).abs().max()))
        print(name, "mismatching calls", n, "/", reps, " failing tiles by wave slot:", slots.tolist(), " by j:", jhist.tolist())
