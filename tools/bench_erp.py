"""BASELINE configs 3 and 5: full equirectangular renders of a Ricoh360-like scene.

  python tools/bench_erp.py [--views K] [--H 1024 --W 2048] [--term-eps 1e-5] [--mask]        (config 3, 1 GPU)
  python -m torch.distributed.run --nproc-per-node N ... tools/bench_erp.py --views K          (config 5, N GPUs)

Scene: near_far [0.1, 300], r0 0.05, density_shift -10, envmap 3 x 3840 x 1920, grid [150,172,516]
(configs/EgoNeRF/ricoh/common.txt), 128 coarse + 128 fine samples, synthetic smooth-field weights.
Each image's rays are generated on the device (ego_erp_rays), every rank renders a contiguous block of rows of every
view (model replicated, no data-path collective), per-image PSNR against a reference image is reduced with one
2-double all-reduce (renderer.py:156-157 semantics).  The reference image here is the same scene rendered with the
fp32-MFMA arithmetic and no skipping, so the PSNR column measures what the skipping / precision options cost.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth  # noqa: E402
from egonerf_amd.renderer import erp_rays, psnr_from_sse, shard_bounds, volume_renderer  # noqa: E402
from egonerf_amd.synth import build_model as make_model  # noqa: E402


def pose(k: int, K: int) -> np.ndarray:
    a = 2 * np.pi * k / max(K, 1)
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s, 0.3 * c], [0, 1, 0, 0.05 * k], [-s, 0, c, 0.3 * s]], np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--H", type=int, default=1024)
    ap.add_argument("--W", type=int, default=2048)
    ap.add_argument("--chunk", type=int, default=65536)
    ap.add_argument("--term-eps", type=float, default=0.0)
    ap.add_argument("--mask", action="store_true")
    ap.add_argument("--no-reference", action="store_true", help="skip the fp32 / unskipped reference render (no PSNR)")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    cfg = synth.SceneConfig(**synth.RICOH)
    model = make_model(cfg, synth.make_weights(cfg, seed=1234), dev)
    kw = dict(chunk=a.chunk, n_coarse=128, n_fine=128, exp_sampling=True, resampling=True, use_coarse_sample=True, device=dev,
              keep_alpha=False)  # an image render reads rgb only (renderer.py:125-157)
    row0, row1 = shard_bounds(a.H, world, rank)  # contiguous block of rows per rank

    def render(k):
        rays = erp_rays(a.H, a.W, pose(k, a.views), dev, row0, row1 - row0)
        return volume_renderer(rays, model, **kw)[0]

    refs = []
    if not a.no_reference:
        model.mlp_precision = "f32"
        with torch.no_grad():
            refs = [render(k) for k in range(a.views)]
    model.mlp_precision = "f16x3"
    if a.mask:
        model.updateAlphaMask()
        model.use_alpha_mask = True
    model.early_termination_eps = a.term_eps
    with torch.no_grad():
        render(0)  # warm-up
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        imgs = [render(k) for k in range(a.views)]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
    psnrs = []
    for k in range(len(refs)):
        d = imgs[k].double() - refs[k].double()
        stat = torch.stack([(d * d).sum(), torch.tensor(float(d.numel()), device=dev, dtype=torch.float64)])
        if world > 1:
            dist.all_reduce(stat)
        psnrs.append(psnr_from_sse(max(stat[0].item(), 1e-300), stat[1].item()))
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        n_rays = a.views * a.H * a.W
        print(json.dumps(dict(config=f"ERP {a.H}x{a.W}, {a.views} views, 128+128 samples, envmap on, Ricoh-like scene",
                              n_gpus=world, s_per_image=float(t) / a.views, rays_per_s=n_rays / float(t),
                              term_eps=a.term_eps, alpha_mask=a.mask, psnr_vs_f32_unskipped_db=psnrs)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
