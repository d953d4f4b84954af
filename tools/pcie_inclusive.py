"""PCIe-inclusive rate of the headline workload, for DESIGN 6 (never `value`: bench.py times rays resident in HBM).

The reference's volume_renderer (renderer.py:11-79) is handed the ray list on the HOST, moves each 4096-ray chunk to the device
(`rays[...].to(device)`, :26) and, with `empty_gpu_cache`, copies every output of the chunk back (`.cpu().numpy()`, :39-53) -
including the [N, S] alpha that only the entropy loss reads.  Same calls here, through egonerf_amd.renderer.volume_renderer:

    python tools/pcie_inclusive.py            -> one JSON line (M rays/s at 4096 x 512 per hand-over form)
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth  # noqa: E402
from egonerf_amd.renderer import volume_renderer  # noqa: E402

dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
kw = dict(n_coarse=512, n_fine=0, exp_sampling=True, resampling=False, use_coarse_sample=True, chunk=4096, device=dev)
N_CHUNKS = 64
host = torch.from_numpy(synth.make_rays(4096 * N_CHUNKS, seed=1))          # pageable, as a dataset hands it over
pinned = host.pin_memory()
resident = host.to(dev)


def rate(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return round(host.shape[0] / best / 1e6, 3)


out = {"workload": "4096-ray chunks x 512 samples, %d chunks per call, volume_renderer" % N_CHUNKS}
with torch.no_grad():
    volume_renderer(resident, model, keep_alpha=False, **kw)   # warm-up (packs, pools, first-call allocations)
    out["resident_in_resident_out_no_alpha"] = rate(lambda: volume_renderer(resident, model, keep_alpha=False, **kw))
    out["resident_in_resident_out_alpha"] = rate(lambda: volume_renderer(resident, model, keep_alpha=True, **kw))
    out["host_in_resident_out_no_alpha"] = rate(lambda: volume_renderer(host, model, keep_alpha=False, **kw))
    out["pinned_in_resident_out_no_alpha"] = rate(lambda: volume_renderer(pinned, model, keep_alpha=False, **kw))
    out["host_in_host_out_no_alpha"] = rate(lambda: volume_renderer(host, model, keep_alpha=False, empty_gpu_cache=True, **kw))
    out["host_in_host_out_alpha_as_the_reference"] = rate(lambda: volume_renderer(host, model, keep_alpha=True, empty_gpu_cache=True, **kw))
out["unit"] = "M rays/s"
print(json.dumps(out))
