#!/bin/bash
# Longer soak of the training step (round 6, after the walk took over d(basis) and dv): 300 repetitions of the full-size step's 32 gradients, bit
# for bit, with the side stream on (the shipped schedule) - and the same for a head of another shape over the shipped tables (sorted route).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python - > gpurun_out/soak6_train_long.txt 2>&1 <<'PY'
import numpy as np, torch
from egonerf_amd import synth
def soak(name, reps, **shape):
    cfg = synth.SceneConfig(**shape)
    model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), "cuda"); model.train()
    N = 8192
    rays = torch.from_numpy(synth.make_rays(N, seed=1)).cuda()
    gt = torch.from_numpy(synth.hash_uniform(3, 0, N * 3).reshape(N, 3).astype(np.float32)).cuda()
    jit = torch.from_numpy(synth.hash_uniform(5, 0, N * 128).reshape(N, 128).astype(np.float32)).cuda()
    u = torch.from_numpy(synth.hash_uniform(5, 1, N * 128).reshape(N, 128).astype(np.float32)).cuda()
    kw = dict(is_train=True, n_coarse=128, n_fine=128, exp_sampling=True, resampling=True, use_coarse_sample=True, jitter=jit, u=u)
    def grads():
        model.zero_grad(set_to_none=True)
        rgb, *_ = model(rays, **kw)
        torch.mean((rgb - gt) ** 2).backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    ref = grads()
    bad = 0
    for rep in range(reps):
        g = grads()
        bad += sum(0 if torch.equal(g[k], ref[k]) else 1 for k in ref)
    finite = all(bool(torch.isfinite(v).all()) for v in ref.values())
    print(f"{name}: training step 8192 x (128+128), {len(ref)} gradient tensors, {reps} repetitions vs the first: {bad} tensors differed in any bit; all finite: {finite}")
soak("shipped shape (tuned path, walk with d(basis) and dv)", 300)
soak("shadingMode MLP over the shipped tables (any-shape head, sorted scatter)", 100, shadingMode="MLP")
PY
cat gpurun_out/soak6_train_long.txt | tail -3
