"""One training step of a shape that takes the any-shape kernels (csrc/ego_generic.hip), three times: the workload of tools/generic_profile.sh."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
dev="cuda"
SHAPES = {"mlp": dict(shadingMode="MLP"), "small": dict(density_n_comp=(8, 8, 8), app_n_comp=(24, 24, 24), featureC=64),
          "enc66": dict(view_pe=6, fea_pe=6, app_dim=12)}
cfg = synth.SceneConfig(**SHAPES[os.environ.get("GEN_SHAPE", "mlp")])
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
model.train()
rays8 = torch.from_numpy(synth.make_rays(8192, seed=1)).to(dev); gt = torch.rand(8192, 3, device=dev)
for _ in range(3):
    model.zero_grad(set_to_none=True)
    rgb, *_ = model(rays8, is_train=True, n_coarse=128, n_fine=128, exp_sampling=True, resampling=True, use_coarse_sample=True)
    torch.mean((rgb - gt) ** 2).backward()
torch.cuda.synchronize()
