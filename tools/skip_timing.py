"""ms per 4096 x 512 batch on the synthetic field at several opacities (density_shift), for the exact skipping modes:
full (every sample: EGO_EXACT_SKIP=0 semantics), tile skip (default), tile skip + need_alpha=False (the march stops at transmittance 0)."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth

dev = torch.device("cuda", 0)
rays = torch.from_numpy(synth.make_rays(4096, seed=1)).to(dev)
out = {}
for shift in [float(x) for x in (sys.argv[1:] or ["-8", "-4", "0", "4"])]:
    cfg = synth.SceneConfig(density_shift=shift)
    model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)

    def timeit(**kw):
        with torch.no_grad():
            for _ in range(400):
                model(rays, n_coarse=512, exp_sampling=True, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                model(rays, n_coarse=512, exp_sampling=True, **kw)
            e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / 100, 4)
    res = {}
    model.skip_zero_weight_tiles = False
    res["full"] = timeit()
    model.skip_zero_weight_tiles = True
    res["tile_skip"] = timeit()
    res["tile_skip_no_alpha"] = timeit(need_alpha=False)
    out[shift] = res
print(json.dumps(out))
