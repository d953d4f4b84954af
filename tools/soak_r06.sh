#!/bin/bash
# Final-build soak (round 6) on the GPU box: determinism of every forward op (3000 repetitions + the full-size batches, all split arithmetics)
# and of the training step's 32 gradients (walk scatter with fixed-point lines + ordered weight-gradient sums).  -> gpurun_out/soak6_*.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 1500 python tools/determinism_check.py 3000 --full > gpurun_out/soak6_determinism.txt 2>&1; tail -6 gpurun_out/soak6_determinism.txt
# (the randomised parity campaign - 4 seeds x 160 cases x 3 arithmetics - is part of the -m gpu suite since round 6)
# round 6: the training step's gradients, bit for bit, 20 repetitions at full size (sorted scatter + ordered weight-gradient sums)
timeout 600 python - > gpurun_out/soak6_train_determinism.txt 2>&1 <<'PY'
import numpy as np, torch
from egonerf_amd import synth
cfg = synth.SceneConfig()
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
    return {k: p.grad.clone() for k, p in model.named_parameters()}
ref = grads()
bad = 0
for rep in range(20):
    g = grads()
    bad += sum(0 if torch.equal(g[k], ref[k]) else 1 for k in ref)
print(f"training step 8192 x (128+128), 32 gradient tensors, 20 repetitions vs the first: {bad} tensors differed in any bit")
PY
tail -1 gpurun_out/soak6_train_determinism.txt
