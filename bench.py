#!/usr/bin/env python
"""Headline benchmark: rays/s of EgoNeRF's volume-rendering hot path at BASELINE config 2
(OmniBlender-barbershop shape: grid [150,172,516], 4096-ray batch, 512 samples/ray, eval, no resampling).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one EgoNeRF.forward over one 4096-ray batch already resident in HBM (ego_render_forward:
march/density -> shade (app gather + basis + PE + MLP on the fp32 matrix cores) -> composite).
Rank 0 prints ONE JSON line.  Rays are independent, so ranks shard work with no data-path collective
("scaling": "weak": every rank renders its own batch per step).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from egonerf_amd import synth  # noqa: E402

N_RAYS, N_SAMPLES = 4096, 512
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA dense peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak
# algorithmic bytes / flops per sample: density 3*(4+2) taps * 16 ch * 4 B, appearance ... * 48 ch
B_DENSITY, B_APP = 1152, 3456
FLOP_SAMPLE_SHADE = 2 * (150 * 128 + 128 * 128 + 128 * 3) + 2 * 144 * 27  # MLP + basis


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 0.7 ms each: amortises the barrier + synchronize bracket
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=1024, help="rays in the bounded CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(cfg, weights, n_rays):
    """The oracle (= CPU restatement of the reference's PyTorch path, kind 'port') on this box's host cores.
    ATen's intra-op threading does not scale to hundreds of cores on these small ops, so the thread count is
    picked by a short calibration (the best one is used and reported as `cores`)."""
    from oracle.egonerf_oracle import OracleScene
    logical = os.cpu_count() or 1
    sc = OracleScene(cfg, weights)
    rays = torch.from_numpy(synth.make_rays(n_rays, seed=1))
    cal = {}
    with torch.no_grad():
        for th in sorted({t for t in (8, 16, 32, 64, 128) if t <= logical} | {min(8, logical)}):
            torch.set_num_threads(th)
            sc.forward(rays[:64], n_coarse=N_SAMPLES)
            t = time.perf_counter()
            sc.forward(rays[:128], n_coarse=N_SAMPLES)
            cal[th] = 128 / (time.perf_counter() - t)
        best_th = max(cal, key=cal.get)
        torch.set_num_threads(best_th)
        best = float("inf")
        for _ in range(2):
            t = time.perf_counter()
            out = sc.forward(rays, n_coarse=N_SAMPLES)
            best = min(best, time.perf_counter() - t)
    return dict(value=n_rays / best, unit="rays/s", cores=best_th, kind="port",
                sample=f"{n_rays} rays x {N_SAMPLES} samples, eval, no resampling, best of 2 ({best:.2f} s) with "
                       f"{best_th} ATen threads (calibration rays/s by thread count: "
                       f"{ {k: round(v) for k, v in cal.items()} }; {logical} logical cores; torch {torch.__version__} CPU)"), out, rays


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    # EGO_BENCH_TEST_SHARED_GPU=1: dry-run of the multi-rank path on a box with one GPU (all ranks on cuda:0, gloo)
    shared = os.environ.get("EGO_BENCH_TEST_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL on ROCm

    from egonerf_amd.synth import build_model as make_model
    cfg = synth.SceneConfig()
    weights = synth.make_weights(cfg, seed=1234)
    model = make_model(cfg, weights, dev)
    rays = torch.from_numpy(synth.make_rays(N_RAYS, seed=1 + rank)).to(dev)
    kw = dict(n_coarse=N_SAMPLES, exp_sampling=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(a.warmup):
            out = model(rays, **kw)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = model(rays, **kw)
        barrier()
        dt = time.perf_counter() - t0
    t = torch.tensor([dt], device="cpu" if shared else dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # MAX over ranks of the barrier-bracketed wall time
    dt = float(t.item())

    if rank == 0:
        # ---- per-kernel durations, measured live with events on the launch stream (same work as a step) ----
        from egonerf_amd import _lib
        lib, st = _lib.load(), _lib.stream_handle()
        sc = model.scene()
        M = N_RAYS * N_SAMPLES
        sched = model._sched(N_SAMPLES, dev)
        z = torch.empty(N_RAYS, N_SAMPLES, device=dev)
        alpha, w = torch.empty_like(z), torch.empty_like(z)
        bg = torch.empty(N_RAYS, device=dev)
        rgb = torch.empty(N_RAYS, N_SAMPLES, 3, device=dev)
        crd = torch.empty(N_RAYS, N_SAMPLES, 4, device=dev)
        rgb_map, depth = torch.empty(N_RAYS, 3, device=dev), torch.empty(N_RAYS, device=dev)
        reps = max(a.steps, 5)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
        for i in range(reps + 2):
            e = ev[max(i - 2, 0)]
            e[0].record()
            _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N_RAYS, N_SAMPLES, None, sched.data_ptr(), None, cfg.near, 0,
                                             z.data_ptr(), alpha.data_ptr(), 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march")
            e[1].record()
            _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N_RAYS, N_SAMPLES, rgb.data_ptr(), None, None, st), "shade")
            e[2].record()
            _lib.check(lib.ego_composite(sc, rays.data_ptr(), z.data_ptr(), w.data_ptr(), bg.data_ptr(), rgb.data_ptr(), N_RAYS,
                                         N_SAMPLES, rgb_map.data_ptr(), depth.data_ptr(), None, None, None, st), "composite")
            e[3].record()
        torch.cuda.synchronize()
        ms = np.array([[e[k].elapsed_time(e[k + 1]) for k in range(3)] for e in ev]).mean(0)
        t_march, t_shade, t_comp = (float(x) * 1e-3 for x in ms)
        shade_bytes = (B_APP + 16 + 12) * M         # gathered taps + 16 B coords read + 12 B rgb write, per sample
        shade_gbps = shade_bytes / t_shade / 1e9
        shade_tflops = FLOP_SAMPLE_SHADE * M / t_shade / 1e12
        prec = model.mlp_precision
        kname = "k_shade_h<SHADE>" if prec == "f16x3" else "k_shade<SHADE>"
        # HBM bytes per launch from the committed PMC passes of this build (bench.py cannot collect PMC counters live)
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", "r01", "pmc_traffic.json")))
            if kname in pmc:
                traffic, traffic_src = pmc[kname]["traffic_bytes"], "profiles/r01/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
        except OSError:
            pass
        # matrix-pipe view: f16x3 executes 3 fp16 MFMA flops per algorithmic flop (dense fp16 peak ~2.5 PF)
        mfma = (dict(mode="f16x3", achieved_algorithmic=shade_tflops, executed=3 * shade_tflops, peak=MFMA_F16_PEAK_TFLOPS,
                     unit="TFLOP/s", frac=3 * shade_tflops / MFMA_F16_PEAK_TFLOPS) if prec == "f16x3" else
                dict(mode="f32", achieved=shade_tflops, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=shade_tflops / MFMA_F32_PEAK_TFLOPS))
        l1_peak = 256 * 64 * 2.4  # GB/s: 64 B/clk/CU vector-L1 return path x 256 CUs x 2.4 GHz
        roofline = dict(bound="hbm", kernel=kname, achieved=shade_gbps, peak=HBM_PEAK_GBPS, unit="GB/s",
                        frac=shade_gbps / HBM_PEAK_GBPS, traffic=traffic, traffic_source=traffic_src, ms=t_shade * 1e3,
                        algorithmic_bytes_per_launch=shade_bytes,
                        note="the 94 MB table set is L2/Infinity-Cache resident, so algorithmic bytes/s exceeds the HBM peak; "
                             "the binding resource is instruction issue: VALU and MFMA time add up on a CDNA4 SIMD "
                             "(tools/coissue_probe.hip), see `issue`",
                        issue=dict(mfma_per_tile=243, mfma_cycles_per_tile=243 * 32, valu_per_tile=1800, valu_cycles_per_tile=1800 * 4,
                                   tiles_per_simd=M / 32 / 1024, clock_GHz=2.1,
                                   additive_bound_ms=(243 * 32 + 1800 * 4) * (M / 32 / 1024) / 2.1e6,
                                   frac=(243 * 32 + 1800 * 4) * (M / 32 / 1024) / 2.1e6 / (t_shade * 1e3)),
                        l1=dict(achieved=shade_gbps, peak=l1_peak, unit="GB/s", frac=shade_gbps / l1_peak),
                        mfma=mfma,
                        other_kernels_ms=dict(k_march_density=t_march * 1e3, k_composite=t_comp * 1e3),
                        march_density=dict(achieved=(B_DENSITY + 28) * M / t_march / 1e9, unit="GB/s",
                                           frac=(B_DENSITY + 28) * M / t_march / 1e9 / HBM_PEAK_GBPS),
                        path_algorithmic_GBps=(B_DENSITY + B_APP) * M / (t_march + t_shade + t_comp) / 1e9)

        cpu = None
        parity = None
        if not a.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N=1 only
            cpu, ref, cpu_rays = cpu_baseline(cfg, weights, a.cpu_rays)
            with torch.no_grad():
                got = model(cpu_rays.to(dev), **kw)
            err = float((got[0].cpu() - ref[0]).abs().max())
            mse = float(((got[0].cpu() - ref[0]) ** 2).mean())
            parity = dict(max_abs_rgb_err=err, psnr_vs_oracle_db=float(-10 * np.log10(max(mse, 1e-30))),
                          max_abs_depth_err=float((got[1].cpu() - ref[1]).abs().max()), rays=a.cpu_rays)

        rays_per_s = world * N_RAYS * a.steps / dt
        line = dict(metric="rays/sec at 4096-ray batch, 512 samples (EgoNeRF volume-rendering forward)", value=rays_per_s,
                    unit="rays/s", samples_per_s=rays_per_s * N_SAMPLES, n_gpus=world, steps=a.steps, warmup=a.warmup,
                    ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                    dtype="f32 (tables, interpolation, compositing; matrix products as 3x fp16 MFMA with fp32 accumulate)"
                    if model.mlp_precision == "f16x3" else "f32",
                    data="synthetic",
                    config=dict(workload="OmniBlender barbershop shape: grid [150,172,516], 16x3/48x3 comps, MLP_Fea; "
                                         "4096 rays x 512 samples, eval, no resampling (BASELINE configs[1])",
                                rays_per_step_per_gpu=N_RAYS, samples_per_ray=N_SAMPLES, parallelism=f"ray-sharded x{world}"),
                    roofline=roofline, cpu_baseline=cpu, parity=parity,
                    speedup_vs_cpu=None if cpu is None else rays_per_s / cpu["value"])
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
