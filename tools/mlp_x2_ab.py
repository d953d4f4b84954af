"""EXPERIMENT: ego_mlp_fea (f16f6) as shipped (two waves per SIMD, one tile each) against EGO_MLP_X2=1 (one wave per SIMD, two tiles
at a time, weight fragments read once for both): time and bit-equality of the colours."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egonerf_amd import synth, _lib
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), "cuda")
model.mlp_precision = "f16f6"
sc, lib, st = model.scene(), _lib.load(), _lib.stream_handle()
M = 4096 * 512
rng = np.random.default_rng(0)
feat = torch.from_numpy(rng.normal(0, 0.7, (M, 27)).astype(np.float32)).cuda()
d = rng.normal(0, 1, (M, 3)); d = torch.from_numpy((d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)).cuda()
def run(x2):
    if x2: os.environ["EGO_MLP_X2"] = "1"
    else: os.environ.pop("EGO_MLP_X2", None)
    rgb = torch.empty(M, 3, device="cuda")
    fn = lambda: _lib.check(lib.ego_mlp_fea(sc, d.data_ptr(), feat.data_ptr(), M, rgb.data_ptr(), st), "mlp")
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50, rgb
for rnd in range(3):
    ta, ra = run(False); tb, rb = run(True)
    print(f"round {rnd}: shipped {ta:.4f} ms, two tiles per wave {tb:.4f} ms, max |d rgb| {float((ra - rb).abs().max()):.2e}")
