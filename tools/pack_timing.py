import sys, os
sys.path.insert(0, "/root/repo")
import torch, ctypes as C
from egonerf_amd import synth, _lib
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), "cuda")
sc = model.scene(); lib, st = _lib.load(), _lib.stream_handle()
buf = model._packed
def t(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("ego_pack_mlp: %.1f us per call" % t(lambda: _lib.check(lib.ego_pack_mlp(sc, buf.data_ptr(), st), "pack")))
