"""GPU: the multi-process path with the HIP model.  The GPU box has one device, so two ranks share cuda:0 and talk over gloo
(EGO_BENCH_TEST_SHARED_GPU=1 in bench.py); the rank logic, sharding, reductions and the launcher are the ones an 8-GPU RCCL
run uses."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egonerf_amd import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    from egonerf_amd.renderer import sharded_render, volume_renderer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), "cuda:0")
    rays = torch.from_numpy(synth.make_rays(1001, seed=9)).cuda()  # odd count: uneven shards
    gt = torch.from_numpy(synth.hash_uniform(4, 0, 1001 * 3).reshape(1001, 3).astype(np.float32)).cuda()
    fn = lambda r: volume_renderer(r, model, chunk=256, n_coarse=16, n_fine=16, exp_sampling=True, resampling=True, keep_alpha=False)[0]
    with torch.no_grad():
        out = sharded_render(fn, rays, gt, gather_image=True)
    img = out.get("image")
    q.put((rank, out["lo"], out["hi"], out["psnr"], None if img is None else img.cpu()))
    dist.destroy_process_group()


def test_two_rank_hip_sharded_render_is_bit_identical_to_one_process():
    from egonerf_amd.renderer import psnr_from_sse, volume_renderer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    assert [p.exitcode for p in procs] == [0, 0]
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), "cuda:0")
    rays = torch.from_numpy(synth.make_rays(1001, seed=9)).cuda()
    gt = torch.from_numpy(synth.hash_uniform(4, 0, 1001 * 3).reshape(1001, 3).astype(np.float32))
    with torch.no_grad():
        whole = volume_renderer(rays, model, chunk=4096, n_coarse=16, n_fine=16, exp_sampling=True, resampling=True)[0].cpu()
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 501, 501, 1001)
    assert torch.equal(res[0][4], whole)  # rays are independent and the kernels deterministic: shard-concat == whole, bit for bit
    d = whole.double().clamp(0, 1) - gt.double()
    assert abs(res[0][3] - psnr_from_sse(float((d * d).sum()), d.numel())) < 1e-9 and res[0][3] == res[1][3]


def _bench(*args, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher environment re-executes itself through torch.distributed.run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["EGO_BENCH_TEST_SHARED_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 2 * 4096 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-9
    assert d["roofline"]["frac"] <= 1.0 and d["roofline"]["bound"] == "mfma"


def test_bench_default_line_contract():
    d = _bench("--steps", "10", "--warmup", "2", "--cpu-rays", "128")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] <= 1.0 and rf["unit"] == "TFLOP/s" and 2500.0 <= rf["peak"] <= 5000.0
    assert rf["hbm_algorithmic"]["frac"] > 0 and rf["traffic"] is None or rf["traffic"] > 0
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "processes" in cb["sample"]
    assert d["parity"]["max_abs_rgb_err"] <= 1e-4
    assert d["ms_per_step"] >= rf["ms"]  # a step contains the dominant kernel


def test_bench_train_and_erp_configs_run():
    d = _bench("--config", "train", "--steps", "2", "--warmup", "1")
    assert d["config"]["rays_per_step_per_gpu"] == 8192 and d["value"] > 0 and np.isfinite(d["loss_last"])
    d = _bench("--config", "erp", "--steps", "1", "--warmup", "1", "--erp-size", "128", "256")
    assert d["scaling"] == "strong" and d["value"] > 0 and d["psnr_vs_f32_unskipped_db"][0] > 80
    env = {"EGO_BENCH_TEST_SHARED_GPU": "1"}
    d = _bench("--config", "erp", "--gpus", "2", "--steps", "1", "--warmup", "1", "--erp-size", "128", "256", env_extra=env)
    assert d["n_gpus"] == 2 and d["psnr_vs_f32_unskipped_db"][0] > 80
