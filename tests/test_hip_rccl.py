"""GPU: the process-group code path on RCCL itself (torch.distributed backend "nccl" IS RCCL on ROCm).  The GPU box has one device,
so the group has ONE rank - which still loads librccl, creates the communicator on the MI355X and runs every collective the
multi-GPU path issues (all_reduce of float64 device tensors, all_gather / gather of image tiles, broadcast, barrier) through it;
the two-rank tests (tests/test_hip_multirank.py) cover the sharding logic over gloo.  BASELINE configs[4] (8 x MI355X) is this
code with world_size 8."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from egonerf_amd import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene():
    from egonerf_amd.renderer import erp_rays
    cfg = synth.SceneConfig(n_voxel=20 ** 3, use_envmap=True, envmap_res_H=16)
    model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), "cuda:0")
    H, W = 32, 64
    rays = erp_rays(H, W, torch.eye(4)[:3], "cuda:0")
    gt = torch.from_numpy(synth.hash_uniform(4, 0, H * W * 3).reshape(H * W, 3).astype(np.float32)).cuda()
    return model, rays, gt, H, W


def _run(model, rays, gt, H, W):
    from egonerf_amd.renderer import evaluation, sharded_render, volume_renderer
    kw = dict(n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)
    fn = lambda r: volume_renderer(r, model, chunk=512, keep_alpha=False, **kw)[0]
    with torch.no_grad():
        out = sharded_render(fn, rays, gt, gather_image=True)
        ev = evaluation([rays], [gt], (W, H), model, chunk=512, ws_metrics=True, **kw)
    return out["psnr"], out["image"].cpu(), ev


def _worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))  # exactly as bench.py's Ranks
    assert dist.get_backend() == "nccl"
    model, rays, gt, H, W = _scene()
    res = _run(model, rays, gt, H, W)
    # the raw collectives on device tensors, float64 included (PSNR statistics are float64)
    t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    b = torch.arange(4, device="cuda", dtype=torch.float32)
    dist.broadcast(b, src=0)
    dist.barrier()
    torch.cuda.synchronize()
    q.put((res, t.cpu().tolist(), b.cpu().tolist()))
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_the_sharded_render_and_evaluation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    (psnr, image, ev), t, b = q.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert t == [1.5, 2.5] and b == [0.0, 1.0, 2.0, 3.0]
    # same numbers as without a process group (this process has none)
    model, rays, gt, H, W = _scene()
    psnr1, image1, ev1 = _run(model, rays, gt, H, W)
    assert torch.equal(image, image1) and psnr == psnr1
    for a, c in zip(ev, ev1):
        assert a == c
    assert len(ev) == 4 and -1 < ev[1][0] < 1 and np.isfinite(ev[2][0]) and -1 < ev[3][0] < 1   # gt is noise: SSIM ~ 0


def test_bench_erp_under_a_one_rank_rccl_launcher():
    """bench.py launched by torch.distributed.run with ONE rank: Ranks creates the RCCL group (device_id = its GPU), the barrier
    bracket, the MAX-over-ranks all_reduce and the PSNR all_reduce run on device tensors through RCCL."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "EGO_BENCH_TEST_SHARED_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--config", "erp", "--erp-size", "128", "256",
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--full-out", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["psnr_vs_f32_unskipped_db"][0] > 80 and d["process_group"] == "nccl"
