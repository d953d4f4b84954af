"""Debug of the f16f6 arithmetic on the GPU: device packer vs the numpy emulator, and ego_mlp_fea per sample vs float64."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mfma_emulator as em
from egonerf_amd import synth, _lib

dev = torch.device("cuda", 0)
cfg = synth.SceneConfig(n_voxel=20 ** 3)
w = synth.make_weights(cfg, seed=1234)
model = synth.build_model(cfg, w, dev)
model.mlp_precision = "f16f6"
sc = model.scene()
torch.cuda.synchronize()
blob = model._packed.cpu().numpy()
off = 2 * em.PACKED_FLOATS + 9216 + em.F8_FLOATS
got = blob[off:off + em.F6_FLOATS].view(np.uint32)
exp = em.pack_mlp_f6(w).view(np.uint32)
bad = np.nonzero(got != exp)[0]
print("f6 region: mismatching slots", bad.size, "of", exp.size, "first", bad[:10], [(hex(got[i]), hex(exp[i])) for i in bad[:5]])
lib, st = _lib.load(), _lib.stream_handle()
rng = np.random.default_rng(0)
M = 256
feat = rng.normal(0, 0.7, (M, 27)).astype(np.float32)
d = rng.normal(0, 1, (M, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
tf, td = torch.from_numpy(feat).to(dev), torch.from_numpy(d).to(dev)
res = {}
for prec in ("f16x3", "f16f8", "f16f6"):
    model.mlp_precision = prec
    sc = model.scene()
    rgb = torch.empty(M, 3, device=dev)
    _lib.check(lib.ego_mlp_fea(sc, td.data_ptr(), tf.data_ptr(), M, rgb.data_ptr(), st), "mlp")
    torch.cuda.synchronize()
    res[prec] = rgb.cpu().numpy()
for prec in ("f16f8", "f16f6"):
    e = np.abs(res[prec] - res["f16x3"])
    print(prec, "max", e.max(), "rms", np.sqrt((e ** 2).mean()), "worst sample", np.unravel_index(e.argmax(), e.shape))
