#!/usr/bin/env python
"""Benchmarks of EgoNeRF's volume-rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config render|train|erp]

`--config render` (default) is the headline: BASELINE configs[1] — OmniBlender-barbershop shape (grid [150,172,516]),
4096-ray batch, 512 samples/ray, eval, no resampling.  A step = one EgoNeRF.forward over one batch already resident in HBM
(ego_render_forward: march/density -> shade (appearance gather + basis + PE + MLP on the matrix cores) -> composite).
`--config train` is BASELINE configs[3] (8192 rays x (128+128), forward + backward + FusedAdam + coarse-table refresh per step);
`--config erp` is configs[2] / [4] (Ricoh-like scene, 1024 x 2048 equirectangular images, 128+128 samples, envmap on, rows
sharded over the ranks, per-image PSNR all-reduced).

N > 1: one process per GPU over RCCL.  Either the caller launches the ranks (torch.distributed.run sets RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_*), or — when WORLD_SIZE is absent — this script re-executes itself through torch.distributed.run on
127.0.0.1.  Rays are independent, so ranks shard work with no data-path collective.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from egonerf_amd import synth  # noqa: E402

N_RAYS, N_SAMPLES = 4096, 512
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA dense peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak
MFMA_F8_PEAK_TFLOPS = 5000.0   # dense MX-fp8 peak
N_CU = 256
N_SIMD = N_CU * 4
CLK_PER_MFMA = {"f16x3": 32, "f16f8": 32, "f16f6": 32, "f32": 64}   # issue-to-issue cycles of a dependent-free MFMA stream (tools/coissue_probe.hip)
CLK_PER_FP8_MFMA = 61          # 1.9x an fp16 instruction (tools/fp8_mfma_probe.hip)
CLK_PER_FP6_MFMA = 38          # 1.18x an fp16 instruction (tools/fp6_probe.hip)
MFMA_F6_PEAK_TFLOPS = 10000.0  # dense MX-fp6 / fp4 peak
CLK_PER_VALU = 4               # a wave64 VALU instruction occupies its SIMD for 4 cycles (16 lanes x 4)
B_DENSITY = 1152               # algorithmic density tap bytes per sample: 3 * (4 + 2) taps * 16 channels * 4 B
PREC_CODE = {"f16x3": 0, "f32": 1, "f16f8": 2, "f16f6": 3}


def kernel_info(prec: str) -> dict:
    """Per-tile instruction counts and the algorithmic per-sample figures of the shade kernel, exported by the library itself
    (ego_shade_kernel_info computes them from the constants its loops run over), so they cannot go stale against the kernel."""
    import ctypes
    from egonerf_amd import _lib
    out = (ctypes.c_int32 * 8)()
    _lib.check(_lib.load().ego_shade_kernel_info(PREC_CODE[prec], out, 8), "ego_shade_kernel_info")
    keys = ("samples_per_tile", "mfma_per_tile", "fp8_mfma_per_tile", "flop_per_mfma", "flop_per_fp8_mfma", "fp8_cvt_per_tile",
            "flop_per_sample", "app_tap_bytes_per_sample")
    return dict(zip(keys, (int(v) for v in out)))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["render", "train", "erp", "metrics"], default="render")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="render: skip the short train / erp runs that the default line carries as `secondary`")
    ap.add_argument("--cpu-rays", type=int, default=1024, help="rays in the single-process CPU-baseline sample")
    ap.add_argument("--train-reg", action="store_true", help="train: add the Ricoh configs' TV / L1 / ortho / entropy terms")
    ap.add_argument("--train-eager", action="store_true", help="train: time the eager Python loop instead of the replayed hipGraph of the iteration")
    ap.add_argument("--views", type=int, default=None, help="erp: images per step sequence (default = --steps)")
    ap.add_argument("--erp-size", type=int, nargs=2, default=[1024, 2048], metavar=("H", "W"))
    ap.add_argument("--mask", action="store_true", help="erp: build the reference's alpha mask and apply it (TensorBase.forward semantics)")
    ap.add_argument("--carve", action="store_true", help="erp: give the synthetic field real empty space (synth.carve_empty_space: density exactly 0 "
                    "outside two radial shells and inside a phi wedge); with --mask the line then carries mask-off AND mask-on timings")
    ap.add_argument("--term-eps", type=float, default=0.0, help="erp: early-termination threshold on the transmittance")
    ap.add_argument("--density-shift", type=float, default=None, help="render: override the scene's density_shift (-8) to make the synthetic "
                    "field more / less opaque: shows what the exact zero-weight tile skip does on surface-like scenes (not the headline workload)")
    ap.add_argument("--n-voxel", type=float, default=None, help="render: grid size (default 27e6 -> [150,172,516], 94 MB of tables, cache resident); "
                    "216e6 -> [300,346,1036], ~400 MB of tables > the 256 MiB Infinity Cache: the regime in which the gathers really read HBM")
    ap.add_argument("--fresh-rays", type=int, default=0, metavar="B", help="render: B distinct ray batches used round-robin (a different batch every "
                    "step) instead of re-rendering one batch; with --n-voxel / --fresh-rays the lean variant line is printed (no alt precisions)")
    ap.add_argument("--full-out", default=None, metavar="PATH", help="where the FULL record goes (default: bench_full.json next to bench.py, and "
                    "gpurun_out/bench_full.json when that directory exists); stdout carries only the compact line (< 6 KB)")
    ap.add_argument("--shape", type=int, nargs=3, default=None, metavar=("RAYS", "N_COARSE", "N_FINE"), help="render: another batch shape on the "
                    "headline scene, e.g. 256 64 0 (BASELINE configs[0]) or 4096 256 256 (SURVEY 8(d) Config 2's resampling secondary: the path "
                    "every shipped config uses, configs/EgoNeRF/common.txt:15-23); N_FINE > 0 = inverse-CDF resampling, coarse samples kept")
    ap.add_argument("--cpu-worker", nargs=2, type=int, metavar=("N_RAYS", "THREADS"), help=argparse.SUPPRESS)
    a = ap.parse_args()
    dflt = dict(render=(200, 10), train=(20, 3), erp=(4, 1), metrics=(10, 2))[a.config]  # render: 0.7 ms steps, amortise the barrier bracket
    a.steps = dflt[0] if a.steps is None else a.steps
    a.warmup = dflt[1] if a.warmup is None else a.warmup
    return a


# =====================================================================================================
# launch
# =====================================================================================================
def self_launch(a) -> None:
    """`python bench.py --gpus N` without a launcher: re-execute through torch.distributed.run (one rank per GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    sys.exit(subprocess.call(cmd, env=env))


class Ranks:
    """Process-group plumbing shared by the configs: device binding, barrier bracket, max-over-ranks wall time."""

    def __init__(self, a):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={self.world} (launcher and flag disagree)")
        # EGO_BENCH_TEST_SHARED_GPU=1: dry-run of the multi-rank path on a box with one GPU (all ranks on cuda:0, gloo)
        self.shared = os.environ.get("EGO_BENCH_TEST_SHARED_GPU") == "1"
        if self.shared:
            local = 0
        if not torch.cuda.is_available():
            raise SystemExit("bench.py: no HIP device visible (the EgoNeRF hot path has no CPU fallback)")
        if local >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {self.rank} needs GPU {local} but only {torch.cuda.device_count()} are visible")
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        self.dist = None
        # a launcher (torch.distributed.run sets RANK) gets a process group at ANY world size, so that `--nproc-per-node 1` exercises
        # the RCCL code path (device-tensor all_reduce / barrier) on a one-GPU box
        launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))
        if self.world > 1 or launched:
            import torch.distributed as dist
            try:
                if self.shared:
                    dist.init_process_group("gloo")
                else:
                    dist.init_process_group("nccl", device_id=self.dev)  # RCCL on ROCm
                self.dist = dist
            except Exception:
                if self.world > 1:
                    raise
                # one rank: the group is optional (a stray RANK in the environment must not cost the N = 1 line)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        """MAX over ranks of one rank's wall time (the contract's figure); the per-rank spread of the same interval is kept in
        `self.spread` (min / max / every rank's value), so that a straggler is visible in the line the day an 8-GPU node runs this."""
        t = torch.tensor([seconds], device="cpu" if self.shared else self.dev, dtype=torch.float64)
        if self.dist is not None:
            every = [torch.zeros_like(t) for _ in range(self.world)]
            self.dist.all_gather(every, t)
            vals = [float(v.item()) for v in every]
        else:
            vals = [seconds]
        self.spread = dict(min_s=min(vals), max_s=max(vals), per_rank_s=vals)
        return max(vals)

    def rank_step_ms(self, steps: int) -> dict:
        sp = getattr(self, "spread", None)
        if not sp:
            return None
        return dict(min=sp["min_s"] / steps * 1e3, max=sp["max_s"] / steps * 1e3, per_rank=[v / steps * 1e3 for v in sp["per_rank_s"]],
                    note="wall time of the timed region on each rank / steps; `ms_per_step` is the max")

    def same_on_all_ranks(self, values) -> bool:
        """True iff `values` (floats) are bit-identical on every rank (all_gather of float64)."""
        if self.dist is None:
            return True
        t = torch.tensor(list(values), device="cpu" if self.shared else self.dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(every, t)
        return all(bool(torch.equal(e, every[0])) for e in every)

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


RAMP_SECONDS = float(os.environ.get("EGO_BENCH_RAMP_SECONDS", "0.25"))


def timed(rk: Ranks, step, steps: int, warmup: int) -> float:
    """Untimed clock ramp + W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks.
    The ramp (the same step repeated for RAMP_SECONDS, untimed) is there because a freshly idle MI355X takes tens of
    milliseconds of load to reach its sustained clocks: with a 13 ms timed region (20 steps of 0.6 ms) straight after 5
    warm-up steps the figure measures the ramp, not the kernel (0.65 vs 0.58 ms per step)."""
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < RAMP_SECONDS:
        for _ in range(8):
            step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    rk.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    rk.barrier()
    return rk.max_over_ranks(time.perf_counter() - t0)


# =====================================================================================================
# CPU baseline (oracle = CPU restatement of the reference's PyTorch path; kind "port")
# =====================================================================================================
def cpu_worker(n_rays: int, threads: int) -> None:
    """One process of the all-cores baseline: build the scene, signal READY, wait for GO, render n_rays in 128-ray chunks."""
    from oracle.egonerf_oracle import OracleScene
    torch.set_num_threads(threads)
    cfg = synth.SceneConfig()
    sc = OracleScene(cfg, synth.make_weights(cfg, seed=1234))
    rays = torch.from_numpy(synth.make_rays(n_rays, seed=100 + os.getpid() % 1000))
    with torch.no_grad():
        sc.forward(rays[:32], n_coarse=N_SAMPLES)
        print("READY", flush=True)
        sys.stdin.readline()
        t = time.perf_counter()
        for lo in range(0, n_rays, 128):
            sc.forward(rays[lo:lo + 128], n_coarse=N_SAMPLES)
        print("DONE", time.perf_counter() - t, flush=True)


def cpu_baseline_all_cores(threads: int = 8, rays_per_proc: int = 1024, max_procs: int = 32):
    """Rays sharded over logical_cores / `threads` processes of `threads` ATen threads each (ATen's intra-op threading alone
    does not scale these small ops past ~16 cores).  Wall time from a common GO to the last DONE."""
    logical = os.cpu_count() or 1
    procs = max(1, min(max_procs, logical // threads))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(rays_per_proc), str(threads)], env=env,
                           stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(procs)]
    try:
        for p in ps:
            if p.stdout.readline().strip() != "READY":
                raise RuntimeError("cpu baseline worker failed to start")
        t0 = time.perf_counter()
        for p in ps:
            p.stdin.write("GO\n")
            p.stdin.flush()
        per = [float(p.stdout.readline().split()[1]) for p in ps]
        wall = time.perf_counter() - t0
    finally:
        for p in ps:
            try:
                p.stdin.close()
                p.wait(timeout=30)
            except Exception:
                p.kill()
    return dict(value=procs * rays_per_proc / wall, unit="rays/s", cores=procs * threads, kind="port",
                sample=f"{procs} processes x {threads} ATen threads, {rays_per_proc} rays x {N_SAMPLES} samples each (128-ray chunks), eval, "
                       f"no resampling; wall {wall:.2f} s from a common start to the last finish (slowest worker {max(per):.2f} s, "
                       f"fastest {min(per):.2f} s); {logical} logical cores; torch {torch.__version__} CPU")


def cpu_baseline_single(cfg, weights, n_rays):
    """One process, ATen thread count picked by a short calibration (the best one is used and reported as `cores`)."""
    from oracle.egonerf_oracle import OracleScene
    logical = os.cpu_count() or 1
    sc = OracleScene(cfg, weights)
    rays = torch.from_numpy(synth.make_rays(n_rays, seed=1))
    cal = {}
    with torch.no_grad():
        for th in sorted({t for t in (8, 16, 32) if t <= logical} | {min(8, logical)}):
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
                sample=f"{n_rays} rays x {N_SAMPLES} samples, eval, no resampling, best of 2 ({best:.2f} s) with {best_th} ATen threads "
                       f"(calibration rays/s by thread count: { {k: round(v) for k, v in cal.items()} })"), out, rays


# =====================================================================================================
# roofline inputs from the committed PMC passes
# =====================================================================================================
def load_pmc(kname: str, need: str = "SQ_INSTS_MFMA_per_SE"):
    """Per-launch counter means of `kname` from the newest profiles/r*/pmc_traffic.json (rocprofv3 --pmc passes of bench.py's
    own step, tools/profile_gpu.sh + tools/pmc_traffic.py).  bench.py cannot collect PMC counters live; `stale` says whether
    the library sources changed since that profile was taken."""
    from egonerf_amd.build import source_hash
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "pmc_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if kname in pmc and need in pmc[kname]:
            return pmc[kname], os.path.relpath(path, REPO), pmc.get("_source_hash") != source_hash()
    return None, None, None


def shade_roofline(prec: str, t_shade: float, M: int, kname_tag: str = "SHADE"):
    """Roofline object of the dominant kernel (shade: appearance gather + basis + PE + MLP).

    `achieved` / `frac` follow SURVEY 8(d): ALGORITHMIC flops per launch (basis + MLP_Fea, 79 712 flop per sample) / the kernel's
    event-timed duration, against the dense fp16 MFMA datasheet peak (2.5 PFLOP/s; the fp32-input MFMA peak for `f32`).  Next to
    it, clearly labelled: `matrix_pipe_busy` (EXECUTED MFMA work incl. the precision-split overhead, each instruction kind against
    its own dense peak), `traffic` (counter-measured HBM bytes per launch) and `hbm_counter_frac` (traffic / time / 8 TB/s)."""
    info = kernel_info(prec)
    kname = {"f16x3": "k_shade_h<SHADE>", "f16f8": "k_shade_h<SHADE,f16f8>", "f16f6": "k_shade_h<SHADE,f16f6>", "f32": "k_shade<SHADE>"}[prec]
    pmc, src, stale = load_pmc(kname)
    peak = MFMA_F32_PEAK_TFLOPS if prec == "f32" else MFMA_F16_PEAK_TFLOPS
    alg_flop = info["flop_per_sample"] * M
    alg_tflops = alg_flop / t_shade / 1e12
    tap_bytes = (info["app_tap_bytes_per_sample"] + 16 + 12) * M   # gathered taps + 16 B coords read + 12 B rgb write, per sample
    out = dict(bound="mfma", kernel=kname, unit="TFLOP/s", achieved=alg_tflops, peak=peak, frac=alg_tflops / peak, traffic=None,
               ms=t_shade * 1e3, algorithmic_flop_per_launch=alg_flop, flop_per_sample=info["flop_per_sample"],
               definition="achieved = algorithmic flops (basis + MLP_Fea, tensorBase.py:54-78) per launch / event-timed kernel duration; "
                          "peak = dense fp16 MFMA datasheet peak" if prec != "f32" else "achieved = algorithmic flops / duration; peak = fp32-input MFMA peak")
    tiles = M / info["samples_per_tile"]
    if pmc is not None:
        n_se = 32
        mfma = pmc["SQ_INSTS_MFMA_per_SE"] * n_se
        valu = pmc["SQ_INSTS_VALU_per_SE"] * n_se - mfma   # SQ_INSTS_VALU counts the MFMAs too
        n8 = info["fp8_mfma_per_tile"] * tiles
        n16 = mfma - n8
        flops16, flops8 = n16 * info["flop_per_mfma"], n8 * info["flop_per_fp8_mfma"]
        busy = (flops16 / (peak * 1e12) + flops8 / ((MFMA_F6_PEAK_TFLOPS if prec == "f16f6" else MFMA_F8_PEAK_TFLOPS) * 1e12)) / t_shade
        clock_ghz = pmc["GRBM_GUI_ACTIVE"] / (pmc["duration_us"] * 1e3) if "duration_us" in pmc else None
        cvt8 = info["fp8_cvt_per_tile"] * tiles
        clk8 = CLK_PER_FP6_MFMA if prec == "f16f6" else CLK_PER_FP8_MFMA
        cvt_extra = 15 if prec == "f16f6" else 1   # issue slots beyond the first: a 32-value fp6 conversion holds the VALU for 64 clk, an fp8 pair conversion for 8
        bound_clk = (n16 * CLK_PER_MFMA[prec] + n8 * clk8 + (valu + cvt8 * cvt_extra) * CLK_PER_VALU) / N_SIMD
        traffic = pmc.get("traffic_bytes")
        out.update(traffic=traffic,
                   hbm_counter_frac=None if traffic is None else traffic / t_shade / (HBM_PEAK_GBPS * 1e9),
                   matrix_pipe_busy=busy, executed_TFLOPs=(flops16 + flops8) / t_shade / 1e12,
                   inputs=dict(source=src, stale_vs_current_sources=stale, mfma_insts_per_launch=mfma, fp8_mfma_insts_per_launch=n8,
                               valu_insts_per_launch=valu, mfma_per_tile=mfma / tiles, valu_per_tile=valu / tiles,
                               mfma_per_tile_from_library=info["mfma_per_tile"] + info["fp8_mfma_per_tile"],
                               flop_per_mfma=info["flop_per_mfma"], flop_per_fp8_mfma=info["flop_per_fp8_mfma"] or None,
                               effective_clock_GHz=clock_ghz),
                   issue=None if clock_ghz is None else dict(
                       note="VALU + MFMA issue time per SIMD summed (an upper estimate of the issue time, not a hard bound: conversion-class VALU runs beside the matrix pipe, fp32-FMA-class VALU competes with it, tools/agpr_coissue_probe.hip)",
                       clk_per_mfma=CLK_PER_MFMA[prec], clk_per_fp8_mfma=clk8 if n8 else None, clk_per_valu=CLK_PER_VALU,
                       bound_ms=bound_clk / (clock_ghz * 1e6), frac=bound_clk / (clock_ghz * 1e6) / (t_shade * 1e3)))
        if clock_ghz is not None and "TA_BUSY_avr" in pmc and "SQ_INSTS_VMEM_RD_per_SE" in pmc:
            # third resource: the vector L1's address / data path.  A wave64 global_load_dwordx4 occupies it for 16 clk whatever its
            # active lanes (6.8 ns per instruction per CU at any occupancy: tools/l1_exec_probe.hip), so its floor is the load count
            vmem = pmc["SQ_INSTS_VMEM_RD_per_SE"] * n_se
            ta_ms = vmem * 16 / N_CU / (clock_ghz * 1e6)
            out["l1"] = dict(note="vector L1 / texture-addresser path: 16 clk per wave64 dwordx4 load (tools/l1_exec_probe.hip); busy = TA_BUSY_avr / GRBM_GUI_ACTIVE of the counter pass",
                             vmem_rd_insts_per_tile=vmem / tiles, bound_ms=ta_ms, frac=ta_ms / (t_shade * 1e3),
                             ta_busy_frac_counter_pass=pmc["TA_BUSY_avr"] / pmc["GRBM_GUI_ACTIVE"])
    else:
        out["inputs"] = dict(source=None, note="no profiles/r*/pmc_traffic.json entry for " + kname)
    out["hbm_algorithmic_GBps"] = tap_bytes / t_shade / 1e9
    out["note"] = ("SURVEY 8(d)'s algorithmic tap bytes / time (hbm_algorithmic_GBps) exceed the 8 TB/s HBM peak because the 94 MB table set is "
                   "resident in L2 / the 256 MiB Infinity Cache: the north_star's '>= 60 % HBM utilisation on the grid-sample kernel' is met on "
                   "that algorithmic definition and does not apply physically (the counters see `traffic` bytes = hbm_counter_frac of the peak); "
                   "the kernel is bound by SIMD instruction issue (`issue`), so the roofline is quoted against the matrix pipe")
    return out


# =====================================================================================================
# --config render (headline, BASELINE configs[1])
# =====================================================================================================
def run_render(a, rk: Ranks):
    from egonerf_amd import _lib
    dev = rk.dev
    cfg = synth.SceneConfig() if a.density_shift is None else synth.SceneConfig(density_shift=a.density_shift)
    weights = synth.make_weights(cfg, seed=1234)
    model = synth.build_model(cfg, weights, dev)
    rays = torch.from_numpy(synth.make_rays(N_RAYS, seed=1 + rk.rank)).to(dev)
    kw = dict(n_coarse=N_SAMPLES, exp_sampling=True)
    with torch.no_grad():
        dt = timed(rk, lambda: model(rays, **kw), a.steps, a.warmup)
    headline_spread = rk.rank_step_ms(a.steps)
    if rk.rank != 0:
        return None
    # ---- per-kernel durations, measured live with events on the launch stream (same work as a step) ----
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
    reps = max(min(a.steps, 200), 5)
    # ego_render_forward shades and composites in one launch where the scene allows it and whole rays divide evenly over the kernel's waves
    # (ego_render_forward_folds; EGO_RENDER_FOLD=0 keeps two launches); then the shade figure below is that kernel's (compositing included)
    folded = bool(lib.ego_render_forward_folds(sc, N_RAYS, N_SAMPLES))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    for i in range(reps + 2):
        e = ev[max(i - 2, 0)]
        e[0].record()
        _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N_RAYS, N_SAMPLES, None, sched.data_ptr(), None, cfg.near, 0,
                                         z.data_ptr(), alpha.data_ptr(), 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march")
        e[1].record()
        if folded:   # what ego_render_forward launches for this scene: shading with the compositing in its epilogue, no k_composite
            _lib.check(lib.ego_shade_composite(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), w.data_ptr(), bg.data_ptr(), N_RAYS, N_SAMPLES,
                                               None, rgb_map.data_ptr(), depth.data_ptr(), None, None, st), "shade_composite")
            e[2].record()
        else:
            _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N_RAYS, N_SAMPLES, rgb.data_ptr(), None, None, st), "shade")
            e[2].record()
            _lib.check(lib.ego_composite(sc, rays.data_ptr(), z.data_ptr(), w.data_ptr(), bg.data_ptr(), rgb.data_ptr(), N_RAYS,
                                         N_SAMPLES, rgb_map.data_ptr(), depth.data_ptr(), None, None, None, st), "composite")
        e[3].record()
    torch.cuda.synchronize()
    ms = np.array([[e[k].elapsed_time(e[k + 1]) for k in range(3)] for e in ev]).mean(0)
    t_march, t_shade, t_comp = (float(x) * 1e-3 for x in ms)
    roofline = shade_roofline(model.mlp_precision, t_shade, M)
    if folded:   # the instantiation ego_render_forward launches: shading with the compositing in its epilogue
        roofline["kernel"] = roofline["kernel"].replace(">", ",FOLD>")
    march_gbps = (B_DENSITY + 28) * M / t_march / 1e9
    march = dict(hbm_algorithmic_GBps=march_gbps, frac_of_hbm_peak=march_gbps / HBM_PEAK_GBPS,
                 note="cache-resident like the shade gather (24.7 MB of density tables); the binding resource is VALU issue")
    mp, msrc, mstale = load_pmc("k_march_density<16>", need="SQ_INSTS_VALU_per_SE")
    if mp is not None and "duration_us" in mp:
        clk = mp["GRBM_GUI_ACTIVE"] / (mp["duration_us"] * 1e3)
        valu = mp["SQ_INSTS_VALU_per_SE"] * 32
        bound_ms = valu * CLK_PER_VALU / N_SIMD / (clk * 1e6)
        march["issue"] = dict(source=msrc, stale_vs_current_sources=mstale, valu_insts_per_launch=valu, valu_per_64_samples=valu / (M / 64),
                              effective_clock_GHz=clk, bound_ms=bound_ms, frac=bound_ms / (t_march * 1e3))
    # exact tile skipping (always on in EgoNeRF.forward's eval path: 32-sample tiles whose weights are all exactly 0 are not shaded;
    # the per-kernel timings above shade every tile): how much of THIS synthetic batch it skips
    zero_tiles = float((w.view(-1, 32).max(dim=1).values == 0).float().mean()) if M % 32 == 0 else None
    roofline.update(other_kernels_ms=dict(k_march_density=t_march * 1e3, k_composite=t_comp * 1e3),
                    compositing_folded_into_shade=folded,
                    exact_zero_weight_tile_skip=dict(enabled=os.environ.get("EGO_EXACT_SKIP", "1") != "0", tiles_skipped_frac=zero_tiles,
                                                     note="bit-identical outputs (the reference adds w * rgb = 0 for those samples); the synthetic "
                                                          "field is semi-transparent, so almost nothing is skipped here"),
                    march_density=march,
                    path_algorithmic_GBps=(B_DENSITY + kernel_info(model.mlp_precision)["app_tap_bytes_per_sample"]) * M / (t_march + t_shade + t_comp) / 1e9)

    # SURVEY 8(d)'s second figure: the UNIQUE texel footprint of the batch (per ray: distinct taps of its samples in each plane / line
    # of its grid), i.e. what the gathers would read if every ray kept its texels - between the algorithmic tap bytes (every tap of
    # every sample) and the counter-measured HBM traffic (what the caches did not absorb)
    def unique_taps(res):
        flat = crd.view(N_RAYS, N_SAMPLES, 4)
        g = (flat[..., 3] != 0).long()
        idx = []
        for a_, n in enumerate(res):
            ix = ((flat[..., a_] + 1.0) * (0.5 * (n - 1))).floor().long()
            idx.append((ix.clamp(0, n - 1), (ix + 1).clamp(0, n - 1)))

        def distinct(keys):  # [N, K] -> number of distinct keys per ray, summed
            ks, _ = torch.sort(keys, dim=1)
            return int((1 + (ks[:, 1:] != ks[:, :-1]).sum(dim=1)).sum())
        total = 0
        for (ax, ay) in ((0, 1), (0, 2), (1, 2)):   # planes (x, y)
            keys = [(g * res[ay] + y) * res[ax] + x for x in idx[ax] for y in idx[ay]]
            total += distinct(torch.cat(keys, dim=1))
        for al in (2, 1, 0):                         # lines
            total += distinct(torch.cat([g * res[al] + l for l in idx[al]], dim=1))
        return total
    try:
        taps = unique_taps([int(v) for v in model.gridSize.tolist()])
        fp_app, fp_dens = taps * 48 * 4, taps * 16 * 4
        roofline["unique_footprint"] = dict(texels_per_ray=taps / N_RAYS, app_bytes_per_launch=fp_app, density_bytes_per_launch=fp_dens,
                                            shade_GBps=fp_app / t_shade / 1e9, frac_of_hbm_peak=fp_app / t_shade / 1e9 / HBM_PEAK_GBPS,
                                            note="per-ray distinct taps x channel bytes (fp32): SURVEY 8(d)'s unique-footprint figure, "
                                                 "next to hbm_algorithmic (every tap of every sample) and traffic (HBM counters)")
    except Exception as e:  # a reporting extra: never fail the bench line over it
        roofline["unique_footprint"] = dict(error=repr(e))

    # The other arithmetics on the same batch, each with its shade-kernel time AND its step-level time (the whole EgoNeRF.forward, event
    # timed over 50 steps): "f16x3" = fp32-grade three-term fp16 split; "app_f16+f16f8" = the config name's literal "bf16" reading
    # (half-precision appearance tables + f16f8 products).  Errors vs the oracle are filled in below (parity leg).
    main_prec = model.mlp_precision
    alt = {}

    def time_variant(prec, app16):
        model.mlp_precision = prec
        model.app_table_dtype = "f16" if app16 else "f32"
        sc2 = model.scene()
        if folded and not app16 and prec != "f32":
            shade = lambda: _lib.check(lib.ego_shade_composite(sc2, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), w.data_ptr(), bg.data_ptr(), N_RAYS, N_SAMPLES,
                                                               None, rgb_map.data_ptr(), depth.data_ptr(), None, None, st), "shade_composite")
        else:
            shade = lambda: _lib.check(lib.ego_shade(sc2, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N_RAYS, N_SAMPLES, rgb.data_ptr(), None, None, st), "shade")
        t_end = time.perf_counter() + 0.1   # untimed ramp, as for the headline: the host-side work above let the clocks drop
        while time.perf_counter() < t_end:
            for _ in range(8):
                shade()
            torch.cuda.synchronize()
        e2 = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        for i in range(23):
            if i >= 2:
                e2[i - 2].record()
            shade()
        torch.cuda.synchronize()
        shade_ms = float(np.mean([e2[i].elapsed_time(e2[i + 1]) for i in range(20)]))
        with torch.no_grad():
            for _ in range(10):
                model(rays, **kw)
            groups = []   # median of five groups of ten: one hiccup (a 1.7 ms outlier was seen once) must not become the figure
            for _ in range(5):
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                for _ in range(10):
                    model(rays, **kw)
                s1.record()
                torch.cuda.synchronize()
                groups.append(s0.elapsed_time(s1) / 10)
        step_ms = float(np.median(groups))
        return dict(shade_ms=shade_ms, ms_per_step=step_ms, rays_per_s=N_RAYS / (step_ms * 1e-3))

    variants = {"f16x3": ("f16x3", False), "f16f8": ("f16f8", False), "f16f6": ("f16f6", False), "app_f16+f16f8": ("f16f8", True)}
    for name, (prec, app16) in variants.items():
        if name == main_prec:
            continue
        alt[name] = time_variant(prec, app16)
    model.mlp_precision, model.app_table_dtype = main_prec, "f32"
    roofline["alt_precision"] = alt

    # opt-in lossy early termination (model.early_termination_eps; not the headline: EgoNeRF.forward shades every sample and the
    # default path only skips what is exactly zero): same step with weights behind transmittance < 1e-5 dropped, and what it costs
    with torch.no_grad():
        exact = model(rays, **kw)
        model.early_termination_eps = 1e-5
        for _ in range(10):
            lossy = model(rays, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            model(rays, **kw)
        e1.record()
        torch.cuda.synchronize()
        model.early_termination_eps = 0.0
    roofline["alt_early_termination"] = dict(eps=1e-5, ms_per_step=e0.elapsed_time(e1) / 50, max_abs_rgb_diff_vs_default=float((lossy[0] - exact[0]).abs().max()),
                                             note="opt-in (model.early_termination_eps), not part of `value`")

    cpu = cpu_single = parity = None
    if not a.no_cpu_baseline and rk.world == 1:  # the CPU baseline is timed at N=1 only
        cpu_single, ref, cpu_rays = cpu_baseline_single(cfg, weights, a.cpu_rays)
        with torch.no_grad():
            got = model(cpu_rays.to(dev), **kw)
        err = float((got[0].cpu() - ref[0]).abs().max())
        mse = float(((got[0].cpu() - ref[0]) ** 2).mean())
        dp, p_hip, p_ref = synth.delta_psnr(got[0].cpu().clamp(0, 1).numpy(), ref[0].clamp(0, 1).numpy())
        parity = dict(max_abs_rgb_err=err, psnr_vs_oracle_db=float(-10 * np.log10(max(mse, 1e-30))),
                      delta_psnr_db=dp, psnr_hip_vs_gt_db=p_hip, psnr_ref_vs_gt_db=p_ref,   # north_star: within 1e-3 dB on a ~30 dB target (synth.psnr_target)
                      max_abs_depth_err=float((got[1].cpu() - ref[1]).abs().max()), rays=a.cpu_rays, mlp_precision=main_prec,
                      tolerance=dict(rgb=1e-4, depth=1e-3 * 23.3, delta_psnr_db=1e-3))
        for name in alt:
            model.mlp_precision, model.app_table_dtype = variants[name][0], ("f16" if variants[name][1] else "f32")
            with torch.no_grad():
                g2 = model(cpu_rays.to(dev), **kw)
            alt[name]["max_abs_rgb_err"] = float((g2[0].cpu() - ref[0]).abs().max())
        model.mlp_precision, model.app_table_dtype = main_prec, "f32"
        try:
            cpu = cpu_baseline_all_cores()
        except Exception as e:  # the single-process figure still stands
            cpu = dict(cpu_single, note=f"all-cores run failed: {e!r}")
    rays_per_s = rk.world * N_RAYS * a.steps / dt
    return dict(metric="rays/sec at 4096-ray batch, 512 samples (EgoNeRF volume-rendering forward)", value=rays_per_s,
                unit="rays/s", samples_per_s=rays_per_s * N_SAMPLES, n_gpus=rk.world, steps=a.steps, warmup=a.warmup,
                rank_step_ms=headline_spread,
                clock_ramp_s=RAMP_SECONDS,  # untimed: the same step repeated before the W warm-up steps (see timed())
                ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype={"f16x3": "f32 (tables, interpolation, compositing; matrix products as 3x fp16 MFMA with fp32 accumulate)",
                       "f16f8": "f32 (tables, interpolation, compositing; matrix products: fp16 MFMA main term + block-scaled fp8 MFMA "
                                "correction terms in the MLP, 3x fp16 MFMA in the basis, fp32 accumulate)",
                       "f16f6": "f32 (tables, interpolation, compositing; matrix products: fp16 MFMA main term + block-scaled fp6 MFMA "
                                "correction terms with per-lane dynamic block scales in the MLP, 3x fp16 MFMA in the basis, fp32 accumulate)",
                       "f32": "f32"}[model.mlp_precision],
                data="synthetic",
                config=dict(workload="OmniBlender barbershop shape: grid [150,172,516], 16x3/48x3 comps, MLP_Fea; "
                                     "4096 rays x 512 samples, eval, no resampling (BASELINE configs[1])"
                                     + ("" if a.density_shift is None else f"; NON-HEADLINE variant: density_shift {a.density_shift}"),
                            rays_per_step_per_gpu=N_RAYS, samples_per_ray=N_SAMPLES, parallelism=f"ray-sharded x{rk.world}"),
                roofline=roofline, cpu_baseline=cpu, cpu_baseline_single_process=cpu_single, parity=parity,
                speedup_vs_cpu=None if cpu is None else rays_per_s / cpu["value"])



# =====================================================================================================
# render variants: fresh ray batches every step / a table set larger than the Infinity Cache
# =====================================================================================================
def run_render_variant(a, rk: Ranks):
    """The headline step (4096 rays x 512 samples, eval, no resampling) with (a) `--fresh-rays B`: B distinct ray batches used
    round-robin, so that no step re-reads the texels the previous one left in the caches, and / or (b) `--n-voxel V`: another grid
    size - 216e6 gives ~400 MB of tables, more than the 256 MiB Infinity Cache, the only regime in which SURVEY 8(d)'s byte
    roofline can bind physically.  The roofline object is the HBM one: achieved = ALGORITHMIC tap bytes of march + shade per step /
    their event-timed duration; `traffic` = counter HBM bytes of the same two kernels when profiles/r*/pmc_traffic.json has a
    section for this variant (tools/profile_r04.sh)."""
    from egonerf_amd import _lib
    dev = rk.dev
    cfg = synth.SceneConfig() if not a.n_voxel else synth.SceneConfig(n_voxel=float(a.n_voxel))
    weights = synth.make_weights(cfg, seed=1234)
    model = synth.build_model(cfg, weights, dev)
    B = max(int(a.fresh_rays), 1)
    batches = [torch.from_numpy(synth.make_rays(N_RAYS, seed=1000 * rk.rank + 1 + b)).to(dev) for b in range(B)]
    kw = dict(n_coarse=N_SAMPLES, exp_sampling=True)
    state = dict(i=0)

    def step():
        model(batches[state["i"] % B], **kw)
        state["i"] += 1
    with torch.no_grad():
        dt = timed(rk, step, a.steps, a.warmup)
    spread = rk.rank_step_ms(a.steps)
    if rk.rank != 0:
        return None
    lib, st = _lib.load(), _lib.stream_handle()
    sc = model.scene()
    M = N_RAYS * N_SAMPLES
    sched = model._sched(N_SAMPLES, dev)
    z = torch.empty(N_RAYS, N_SAMPLES, device=dev)
    w = torch.empty_like(z)
    bg = torch.empty(N_RAYS, device=dev)
    rgb = torch.empty(N_RAYS, N_SAMPLES, 3, device=dev)
    crd = torch.empty(N_RAYS, N_SAMPLES, 4, device=dev)
    rgb_map, depth = torch.empty(N_RAYS, 3, device=dev), torch.empty(N_RAYS, device=dev)
    reps = max(min(a.steps, 128), B, 8)
    # ego_render_forward shades and composites in one launch where the scene allows it and whole rays divide evenly over the kernel's waves
    # (ego_render_forward_folds; EGO_RENDER_FOLD=0 keeps two launches); then the shade figure below is that kernel's (compositing included)
    folded = bool(lib.ego_render_forward_folds(sc, N_RAYS, N_SAMPLES))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    for i in range(reps + 2):
        e, rays = ev[max(i - 2, 0)], batches[i % B]
        e[0].record()
        _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N_RAYS, N_SAMPLES, None, sched.data_ptr(), None, cfg.near, 0,
                                         z.data_ptr(), None, 0, w.data_ptr(), bg.data_ptr(), crd.data_ptr(), None, None, st), "march")
        e[1].record()
        if folded:   # what ego_render_forward launches for this scene: shading with the compositing in its epilogue, no k_composite
            _lib.check(lib.ego_shade_composite(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), w.data_ptr(), bg.data_ptr(), N_RAYS, N_SAMPLES,
                                               None, rgb_map.data_ptr(), depth.data_ptr(), None, None, st), "shade_composite")
            e[2].record()
        else:
            _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N_RAYS, N_SAMPLES, rgb.data_ptr(), None, None, st), "shade")
            e[2].record()
            _lib.check(lib.ego_composite(sc, rays.data_ptr(), z.data_ptr(), w.data_ptr(), bg.data_ptr(), rgb.data_ptr(), N_RAYS,
                                         N_SAMPLES, rgb_map.data_ptr(), depth.data_ptr(), None, None, None, st), "composite")
        e[3].record()
    torch.cuda.synchronize()
    ms = np.array([[e[k].elapsed_time(e[k + 1]) for k in range(3)] for e in ev]).mean(0)
    t_march, t_shade, t_comp = (float(x) * 1e-3 for x in ms)
    info = kernel_info(model.mlp_precision)
    table_bytes = sum(int(p.numel()) * 4 for n, p in model.named_parameters() if "plane" in n or "line" in n)
    alg_shade = (info["app_tap_bytes_per_sample"] + 16 + 12) * M
    alg_march = (B_DENSITY + 28) * M
    key = "big_grid" if a.n_voxel else ("fresh_rays" if B > 1 else None)
    pmc, src, stale = load_pmc_section(key) if key else (None, None, None)
    t_gs = t_march + t_shade
    roofline = dict(bound="hbm", kernel="k_march_density + k_shade_h (the two grid-sample kernels)", unit="GB/s", peak=HBM_PEAK_GBPS,
                    achieved=None, frac=None, algorithmic_GBps=(alg_shade + alg_march) / t_gs / 1e9,
                    algorithmic_frac=(alg_shade + alg_march) / t_gs / 1e9 / HBM_PEAK_GBPS,
                    algorithmic_bytes_per_step=alg_shade + alg_march, traffic=None, hbm_counter_GBps=None, hbm_counter_frac=None,
                    kernels_ms=dict(k_march_density=t_march * 1e3, k_shade=t_shade * 1e3, k_composite=t_comp * 1e3),
                    shade=dict(algorithmic_GBps=alg_shade / t_shade / 1e9, mfma_frac=info["flop_per_sample"] * M / t_shade / 1e12 / MFMA_F16_PEAK_TFLOPS),
                    march=dict(algorithmic_GBps=alg_march / t_march / 1e9),
                    table_bytes=table_bytes, infinity_cache_bytes=256 * 2 ** 20, tables_fit_infinity_cache=table_bytes < 256 * 2 ** 20,
                    definition="bound 'hbm': achieved / frac = COUNTER bytes (traffic = FETCH_SIZE x 2 + WRITE_SIZE of the two grid-sample kernels, "
                               "separate rocprofv3 --pmc passes, per step) / their event-timed duration, against 8 TB/s - a physical fraction, <= 1; "
                               "null when no counter pass is tracked.  algorithmic_GBps / algorithmic_frac = SURVEY 8(d)'s tap bytes (4 608 B per "
                               "sample + coords / outputs) / the same time: exceeds 1 when the caches, not HBM, serve the taps - not a roofline")
    if pmc is not None and pmc.get("traffic_bytes"):
        roofline.update(traffic=pmc["traffic_bytes"], hbm_counter_GBps=pmc["traffic_bytes"] / t_gs / 1e9,
                        hbm_counter_frac=pmc["traffic_bytes"] / t_gs / 1e9 / HBM_PEAK_GBPS,
                        achieved=pmc["traffic_bytes"] / t_gs / 1e9, frac=pmc["traffic_bytes"] / t_gs / 1e9 / HBM_PEAK_GBPS,
                        inputs=dict(source=src, stale_vs_current_sources=stale, per_kernel=pmc.get("per_kernel")))
    rays_per_s = rk.world * N_RAYS * a.steps / dt
    return dict(metric="rays/sec at 4096-ray batch, 512 samples (EgoNeRF volume-rendering forward)", value=rays_per_s, unit="rays/s",
                samples_per_s=rays_per_s * N_SAMPLES, n_gpus=rk.world, steps=a.steps, warmup=a.warmup, clock_ramp_s=RAMP_SECONDS,
                ms_per_step=dt / a.steps * 1e3, rank_step_ms=spread, higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype="f32 (tables, interpolation, compositing; matrix products: mlp_precision = " + model.mlp_precision + ")", data="synthetic",
                config=dict(workload=f"grid {cfg.grid} ({table_bytes / 2 ** 20:.0f} MiB of VM tables), 4096 rays x 512 samples, eval, no resampling; "
                                     f"{B} distinct ray batch(es) used round-robin (a different batch every step)" if B > 1 else
                                     f"grid {cfg.grid} ({table_bytes / 2 ** 20:.0f} MiB of VM tables), 4096 rays x 512 samples, eval, no resampling",
                            ray_batches=B, rays_per_step_per_gpu=N_RAYS, samples_per_ray=N_SAMPLES, parallelism=f"ray-sharded x{rk.world}"),
                roofline=roofline, cpu_baseline=None)


def run_render_shape(a, rk: Ranks):
    """`--shape RAYS N_COARSE N_FINE` on the headline scene: BASELINE configs[0] (256 x 64, the reference's own CPU-runnable case) and
    SURVEY 8(d) Config 2's secondary (4096 x (256 + 256), resampling on - what every shipped config runs, configs/EgoNeRF/common.txt:15-23).
    A step = one EgoNeRF.forward over one resident batch; per-kernel split through the stage entry points; parity and cpu_baseline
    against the oracle on the same rays."""
    from egonerf_amd import _lib
    dev = rk.dev
    N, nc, nf = (int(v) for v in a.shape)
    cfg = synth.SceneConfig()
    weights = synth.make_weights(cfg, seed=1234)
    model = synth.build_model(cfg, weights, dev)
    rays = torch.from_numpy(synth.make_rays(N, seed=1 + rk.rank)).to(dev)
    kw = dict(n_coarse=nc, exp_sampling=True) if nf == 0 else dict(n_coarse=nc, n_fine=nf, resampling=True, use_coarse_sample=True, exp_sampling=True)
    with torch.no_grad():
        dt = timed(rk, lambda: model(rays, **kw), a.steps, a.warmup)
    spread = rk.rank_step_ms(a.steps)
    if rk.rank != 0:
        return None
    S = nc + nf
    if nf:
        split, skipped = erp_chunk_split(model, rays, reps=24, ERP_NC=nc, ERP_NF=nf)
    else:
        lib, st, sc = _lib.load(), _lib.stream_handle(), model.scene()
        f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        sched = model._sched(nc, dev)
        z, w, bg, crd, rgb, rgb_map, depth = f(N, S), f(N, S), f(N), f(N, S, 4), f(N, S, 3), f(N, 3), f(N)
        act = torch.empty(N * S // 32 + 1, device=dev, dtype=torch.uint8)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(24)]
        for i in range(26):
            e = evs[max(i - 2, 0)]
            e[0].record()
            _lib.check(lib.ego_march_density(sc, rays.data_ptr(), N, S, None, sched.data_ptr(), None, cfg.near, 0, z.data_ptr(), None, 0, w.data_ptr(),
                                             bg.data_ptr(), crd.data_ptr(), None, act.data_ptr(), st), "march")
            e[1].record()
            _lib.check(lib.ego_shade(sc, rays.data_ptr(), z.data_ptr(), crd.data_ptr(), N, S, rgb.data_ptr(), None, act.data_ptr(), st), "shade")
            e[2].record()
            _lib.check(lib.ego_composite(sc, rays.data_ptr(), z.data_ptr(), w.data_ptr(), bg.data_ptr(), rgb.data_ptr(), N, S, rgb_map.data_ptr(),
                                         depth.data_ptr(), None, None, None, st), "composite")
            e[3].record()
        torch.cuda.synchronize()
        ms = np.array([[e[k].elapsed_time(e[k + 1]) for k in range(3)] for e in evs]).mean(0)
        split = {n: float(v) for n, v in zip(("k_march_density", "k_shade", "k_composite"), ms)}
        skipped = float((act[: N * S // 32] == 0).float().mean()) if N * S >= 32 else 0.0
    roofline = shade_roofline(model.mlp_precision, split["k_shade"] * 1e-3, N * S)
    for k in ("issue", "l1", "inputs", "matrix_pipe_busy", "executed_TFLOPs"):   # derived from the 4096 x 512 launch's counters
        roofline.pop(k, None)
    roofline.update(traffic=None, hbm_counter_frac=None, kernels_ms=split, kernels_ms_sum=float(sum(split.values())),
                    exact_zero_weight_tiles_skipped_frac=skipped,
                    note="dominant kernel = k_shade; kernels_ms = the launches of one step, event timed through the stage entry points; a step of "
                         "this size is partly launch-bound (ms_per_step vs kernels_ms_sum)")
    cpu = parity = None
    if not a.no_cpu_baseline and rk.world == 1:
        from oracle.egonerf_oracle import OracleScene
        threads = min(32, os.cpu_count() or 1)
        torch.set_num_threads(threads)
        n_cpu = min(N, 1024)
        orc, r_cpu = OracleScene(cfg, weights), rays[:n_cpu].cpu()
        okw = {k: v for k, v in kw.items() if k != "exp_sampling"}
        best = float("inf")
        with torch.no_grad():
            for _ in range(2):
                t0 = time.perf_counter()
                ref = orc.forward(r_cpu, **okw)
                best = min(best, time.perf_counter() - t0)
            got = model(rays[:n_cpu], **kw)
        err = float((got[0].cpu() - ref[0]).abs().max())
        dp, p_hip, p_ref = synth.delta_psnr(got[0].cpu().clamp(0, 1).numpy(), ref[0].clamp(0, 1).numpy())
        parity = dict(max_abs_rgb_err=err, delta_psnr_db=dp, psnr_hip_vs_gt_db=p_hip, psnr_ref_vs_gt_db=p_ref,
                      max_abs_depth_err=float((got[1].cpu() - ref[1]).abs().max()), rays=n_cpu, tolerance=dict(rgb=1e-4, delta_psnr_db=1e-3))
        cpu = dict(value=n_cpu / best, unit="rays/s", cores=threads, kind="port",
                   sample=f"oracle render of {n_cpu} rays x ({nc}+{nf}) samples, best of 2 ({best:.2f} s) with {threads} ATen threads")
    rays_per_s = rk.world * N * a.steps / dt
    return dict(metric=f"rays/sec at {N}-ray batch, {nc}+{nf} samples (EgoNeRF volume-rendering forward)", value=rays_per_s, unit="rays/s",
                samples_per_s=rays_per_s * S, n_gpus=rk.world, steps=a.steps, warmup=a.warmup, clock_ramp_s=RAMP_SECONDS,
                ms_per_step=dt / a.steps * 1e3, rank_step_ms=spread, higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype="f32 (tables, interpolation, compositing; matrix products: mlp_precision = " + model.mlp_precision + ")", data="synthetic",
                config=dict(workload=f"OmniBlender barbershop shape: grid {cfg.grid}, {N} rays x ({nc}+{nf}) samples, eval, "
                                     + ("inverse-CDF resampling on (coarse samples kept)" if nf else "no resampling")
                                     + (" (BASELINE configs[0] on the GPU)" if (N, nc, nf) == (256, 64, 0) else ""),
                            rays_per_step_per_gpu=N, samples_per_ray=S, parallelism=f"ray-sharded x{rk.world}"),
                roofline=roofline, cpu_baseline=cpu, parity=parity, speedup_vs_cpu=None if cpu is None else rays_per_s / cpu["value"])


def run_eval_metrics(a, rk: Ranks):
    """SURVEY 8(f) row 1 on the measurement bar of the other rows: the evaluation metrics of renderer.py:153-163 on a device-resident
    1024 x 2048 equirectangular image pair - PSNR (renderer.py:156-157), rgb_ssim (utils.py:104-152) with its latitude-weighted variant
    (extra/ws_ssim.py:12-33) and WS-PSNR.  A step = all four numbers of one image, as `evaluation` asks for them (each returns a host float).
    Dominant kernel = k_rgb_ssim (`ego_rgb_ssim`); HBM roofline on its algorithmic bytes (both images read once, the map written once)."""
    from egonerf_amd import metrics
    dev = rk.dev
    H, W = (int(v) for v in (a.erp_size or [1024, 2048]))
    img = torch.from_numpy(synth.hash_uniform(21, 0, H * W * 3).reshape(H, W, 3).astype(np.float32)).to(dev)
    noise = torch.from_numpy(synth.hash_uniform(21, 1, H * W * 3).reshape(H, W, 3).astype(np.float32)).to(dev)
    gt = (img + (noise - 0.5) * 0.11).clamp(0, 1)   # ~30 dB
    out = {}

    def step():
        out["psnr"] = metrics.psnr(img, gt)
        out["ssim"], out["ws_ssim"] = metrics.ws_ssim(img, gt, 1.0)
        out["ws_psnr"] = metrics.ws_psnr(img, gt, 1.0)

    dt = timed(rk, step, a.steps, a.warmup)
    spread = rk.rank_step_ms(a.steps)
    if rk.rank != 0:
        return None
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for i in range(14):
        e = evs[max(i - 2, 0)]
        e[0].record()
        metrics.rgb_ssim(img, gt, 1.0, return_map=True)
        e[1].record()
    torch.cuda.synchronize()
    t_ssim = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs])) * 1e-3
    Ho, Wo = H - 10, W - 10
    alg = 2 * H * W * 3 * 4 + Ho * Wo * 3 * 4
    roofline = dict(bound="hbm", kernel="k_rgb_ssim", unit="GB/s", achieved=alg / t_ssim / 1e9, peak=HBM_PEAK_GBPS, frac=alg / t_ssim / 1e9 / HBM_PEAK_GBPS,
                    traffic=None, ms=t_ssim * 1e3, algorithmic_bytes=alg,
                    note="event-timed ego_rgb_ssim with the map returned (the form ws_ssim uses, incl. the map's allocation); separable 11-tap 'valid' "
                         "Gaussian windows over five moment images in float64 like the reference (~145 fp64 multiply-adds per output value through "
                         "LDS): bound by that arithmetic, not by the 75 MB it moves - the fraction is stated against HBM all the same")
    cpu = parity = None
    if not a.no_cpu_baseline and rk.world == 1:
        from oracle.egonerf_oracle import rgb_ssim as ref_ssim, psnr as ref_psnr
        hc, wc = H, W   # the whole image: ~1 s of one core
        ac, bc = img[:hc, :wc].cpu(), gt[:hc, :wc].cpu()
        t0 = time.perf_counter()
        s_ref = float(ref_ssim(ac.numpy(), bc.numpy(), 1))
        p_ref = float(ref_psnr(ac, bc))
        t_cpu = time.perf_counter() - t0
        s_hip = metrics.rgb_ssim(img[:hc, :wc].contiguous(), gt[:hc, :wc].contiguous(), 1.0)
        p_hip = metrics.psnr(img[:hc, :wc], gt[:hc, :wc])
        parity = dict(abs_ssim_err=abs(s_hip - s_ref), abs_psnr_err_db=abs(p_hip - p_ref), ssim_ref=s_ref, psnr_ref_db=p_ref, crop=[hc, wc],
                      tolerance=dict(ssim=1e-9, psnr_db=1e-3))
        cpu = dict(value=hc * wc / t_cpu / 1e6, unit="Mpixel/s", cores=1, kind="port",
                   sample=f"oracle rgb_ssim (scipy convolve2d, as utils.py:104-152) + PSNR of the {hc} x {wc} image: {t_cpu:.2f} s on one core")
    mpix = rk.world * H * W * a.steps / dt / 1e6
    return dict(metric=f"evaluation metrics (PSNR, SSIM, WS-PSNR, WS-SSIM) of one {H} x {W} image on the device", value=mpix, unit="Mpixel/s",
                n_gpus=rk.world, steps=a.steps, warmup=a.warmup, clock_ramp_s=RAMP_SECONDS, ms_per_step=dt / a.steps * 1e3, rank_step_ms=spread,
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32 images, f64 sums", data="synthetic",
                config=dict(workload=f"{H} x {W} x 3 equirectangular image pair (~30 dB apart), renderer.py:153-163 + extra/ws_ssim.py", parallelism=f"replicas x{rk.world}"),
                values={k: float(v) for k, v in out.items()}, roofline=roofline, cpu_baseline=cpu, parity=parity,
                speedup_vs_cpu=None if cpu is None else mpix / cpu["value"])


# =====================================================================================================
# --config train (BASELINE configs[3])
# =====================================================================================================
TRAIN_RAYS, TRAIN_NC, TRAIN_NF = 8192, 128, 128


def load_pmc_section(key: str):
    """A whole-step section ("train_step" / "erp_image": HBM bytes summed over every kernel of one step, tools/pmc_traffic.py) of
    the newest profiles/r*/pmc_traffic.json that has one."""
    from egonerf_amd.build import source_hash
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "pmc_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if key in pmc:
            return pmc[key], os.path.relpath(path, REPO), pmc.get("_source_hash") != source_hash()
    return None, None, None


def cpu_baseline_train(cfg, weights, n_rays: int):
    """The oracle's training step (EgoNeRF.forward is_train with 128+128 resampling -> MSE -> autograd backward -> torch.optim.Adam
    over all 32 parameter tensors, train.py:245-330) on the host cores, on a bounded ray sample: rays/s = sample / (fwd + bwd) with
    the Adam step (independent of the ray count) timed once and charged pro rata to a full 8192-ray step."""
    from oracle.egonerf_oracle import OracleScene
    logical = os.cpu_count() or 1
    threads = min(32, logical)
    torch.set_num_threads(threads)
    sc = OracleScene(cfg, weights)
    params = list(sc.w.values())
    for v in params:
        v.requires_grad_(True)
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99))
    rays = torch.from_numpy(synth.make_rays(n_rays, seed=1))
    gt = torch.from_numpy(synth.hash_uniform(3, 0, n_rays * 3).reshape(n_rays, 3).astype(np.float32))
    jit = torch.from_numpy(synth.hash_uniform(5, 0, n_rays * TRAIN_NC).reshape(n_rays, TRAIN_NC).astype(np.float32))
    u = torch.from_numpy(synth.hash_uniform(6, 0, n_rays * TRAIN_NF).reshape(n_rays, TRAIN_NF).astype(np.float32))

    def fwd_bwd(n):
        sc.update_coarse_sigma_grid()
        t0 = time.perf_counter()
        rgb = sc.forward(rays[:n], n_coarse=TRAIN_NC, n_fine=TRAIN_NF, resampling=True, is_train=True, jitter=jit[:n], u=u[:n])[0]
        loss = torch.mean((rgb - gt[:n]) ** 2)
        t1 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        return t1 - t0, time.perf_counter() - t1

    fwd_bwd(min(64, n_rays))          # warm-up (allocator, thread pool)
    tf, tb = fwd_bwd(n_rays)
    t0 = time.perf_counter()
    opt.step()
    t_adam = time.perf_counter() - t0
    step_equiv = (tf + tb) * (TRAIN_RAYS / n_rays) + t_adam
    return dict(value=TRAIN_RAYS / step_equiv, unit="rays/s", cores=threads, kind="port",
                sample=f"oracle training step on {n_rays} rays x ({TRAIN_NC}+{TRAIN_NF}) samples with {threads} ATen threads: forward {tf:.2f} s, "
                       f"backward {tb:.2f} s (scaled x{TRAIN_RAYS // n_rays} to the {TRAIN_RAYS}-ray step) + torch.optim.Adam over 24.7 M parameters "
                       f"{t_adam:.2f} s (once per step) -> {step_equiv:.1f} s per {TRAIN_RAYS}-ray step (survey container, 8 threads: 22.8-27.1 s)",
                forward_s=tf, backward_s=tb, adam_s=t_adam, seconds_per_8192_ray_step=step_equiv)


def run_train(a, rk: Ranks):
    from egonerf_amd import train as ego_train
    from egonerf_amd.losses import TVLoss, ray_entropy_loss
    from egonerf_amd.optim import FusedAdam
    dev, N = rk.dev, TRAIN_RAYS
    cfg = synth.SceneConfig()
    weights = synth.make_weights(cfg, seed=1234)
    model = synth.build_model(cfg, weights, dev)
    model.train()
    rays = torch.from_numpy(synth.make_rays(N, seed=1 + rk.rank)).to(dev)
    gt = torch.from_numpy(synth.hash_uniform(3 + rk.rank, 0, N * 3).reshape(N, 3).astype(np.float32)).to(dev)
    # train.py:176-186; lr_factor = lr_decay_target_ratio ** (1 / n_iters) (train.py:171-174 with opt.py's 0.1 / 30000); the step count
    # and the decay live on the device so that the whole iteration can be captured (egonerf_amd.train.GraphedTrainStep)
    opt = FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), capturable=True, lr_factor=0.1 ** (1 / 30000))
    tv = TVLoss()
    kw = dict(is_train=True, n_coarse=TRAIN_NC, n_fine=TRAIN_NF, exp_sampling=True, resampling=True, use_coarse_sample=True)
    losses = []

    def forward():
        rgb, _, _, _, alpha = model(rays, jitter=torch.rand(N, TRAIN_NC, device=dev), u=torch.rand(N, TRAIN_NF, device=dev), **kw)
        loss = torch.mean((rgb - gt) ** 2)
        if a.train_reg:  # configs/EgoNeRF/ricoh/common.txt:12-13 + opt.py defaults
            loss = loss + 1e-4 * model.vector_comp_diffs() + 8e-5 * model.density_L1() + 0.1 * model.TV_loss_density(tv) \
                + 0.01 * model.TV_loss_app(tv) + 1e-3 * ray_entropy_loss(alpha)
        return loss

    def step():
        loss = forward()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        model.update_coarse_sigma_grid()  # every step when resampling (train.py:356-357)
        losses.append(loss.detach())

    # The measured step: the same iteration captured once as a hipGraph and replayed (eager Python queues ~150 launches per
    # iteration and leaves the device gaps between small kernels); the eager loop is timed next to it (--train-eager: only it).
    graphed, graph_error = None, None
    if not a.train_eager:
        def loss_fn(rgb, tgt, alpha):
            loss = torch.mean((rgb - tgt) ** 2)
            if a.train_reg:
                loss = loss + 1e-4 * model.vector_comp_diffs() + 8e-5 * model.density_L1() + 0.1 * model.TV_loss_density(tv) \
                    + 0.01 * model.TV_loss_app(tv) + 1e-3 * ray_entropy_loss(alpha)
            return loss
        try:
            graphed = ego_train.GraphedTrainStep(model, opt, rays, gt, {k: v for k, v in kw.items() if k != "is_train"}, loss_fn=loss_fn, warmup=2)
        except Exception as e:  # a capture problem must not cost the configuration its number: fall back to the eager loop, say so
            graphed, graph_error = None, repr(e)[:400]
            torch.cuda.synchronize()
    if graphed is not None:
        # the step gets faster as the fit proceeds (the scatters skip zero gradients), so the eager loop is timed before AND after
        # the replays and the two are averaged - unless one of them is a host hiccup (the eager loop is host-driven: on a box whose
        # CPUs were busy with another tenant's work one of the two has been seen at twice its usual time while the replays were
        # unaffected), in which case the other one stands
        losses.append(graphed(rays, gt).clone())
        e0 = timed(rk, step, a.steps, 1)
        dt = timed(rk, lambda: graphed(rays, gt), a.steps, a.warmup)
        spread = rk.rank_step_ms(a.steps)
        losses.append(graphed.loss.clone())
        e1 = timed(rk, step, a.steps, 1)
        dt_eager = min(e0, e1) if max(e0, e1) > 1.3 * min(e0, e1) else 0.5 * (e0 + e1)
    else:
        dt = dt_eager = timed(rk, step, a.steps, a.warmup)
        spread = rk.rank_step_ms(a.steps)
    if rk.rank != 0:
        return None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    loss = forward()
    ev[1].record()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    ev[2].record()
    opt.step()
    model.update_coarse_sigma_grid()
    ev[3].record()
    torch.cuda.synchronize()
    # per-call split of one step on ONE stream (the timed steps overlap the two table scatters with the HBM-bound kernels on a side
    # stream, so these intervals sum to more than ms_per_step): events after every library call of the differentiable render
    side = ego_train.SIDE_STREAM_SCATTER
    ego_train.SIDE_STREAM_SCATTER = False
    kernels = {}
    try:
        for rep in range(3):
            marks = ego_train.KERNEL_MARKS = []
            ego_train.mark("begin")
            loss = forward()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            ego_train.mark("autograd_tail")
            opt.step()
            model.update_coarse_sigma_grid()
            ego_train.mark("adam_and_coarse_refresh")
            torch.cuda.synchronize()
            ego_train.KERNEL_MARKS = None
            if rep == 0:
                continue  # first serialised step: allocator warm-up
            seen = {}
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                k = seen[n1] = seen.get(n1, 0) + 1
                name = n1 if n1 not in ("ego_march_density", "ego_weight_grad") else f"{n1}#{k}"
                kernels[name] = kernels.get(name, 0.0) + e0.elapsed_time(e1) / 2
    finally:
        ego_train.KERNEL_MARKS = None
        ego_train.SIDE_STREAM_SCATTER = side
    t_step = dt / a.steps
    rays_per_s = rk.world * N / t_step
    # design traffic of one step: what the kernels exchange through memory by construction (activation dumps written by the forward and
    # read by the shade backward + the weight-gradient passes, gradient activations, coordinates; per fine sample, fp32)
    M = N * (TRAIN_NC + TRAIN_NF)
    # (bytes per fine sample as the kernels exchange them in round 6: h1 / h2 / dh1 / dh2 as halves, ReLU masks as bits, 27 feature slots
    # in 32 floats, dv fp32; the x dump is gone since r05, the scatter reads dv ONCE since r06)
    # r06, late: v is not dumped any more - d(basis) rides along in the appearance scatter's walk, which reads dfe (128 B) once per sort -
    # and dv is not written any more either: the walk re-derives it from those 128 B
    design = dict(forward_dumps_written=(256 + 256 + 32 + 128) * M, shade_bwd_read=(128 + 32 + 16 + 12 + 12) * M,
                  shade_bwd_written=(256 + 256 + 128 + 8) * M, wgrad_read=((256 + 256) + (256 + 128) + (12 + 256)) * M,
                  scatter_read=(3 * 128 + 4 + 6 * (16 + 4)) * M + 3 * 2 * 16 * (M // 13), adam=24_721_123 * 28)
    design_total = float(sum(design.values()))
    pmc, src, stale = load_pmc_section("train_step")
    traffic = None if pmc is None else pmc.get("traffic_bytes")
    roofline = dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBPS, traffic=traffic,
                    achieved=None if traffic is None else traffic / t_step / 1e9,
                    frac=None if traffic is None else traffic / t_step / 1e9 / HBM_PEAK_GBPS,
                    definition="achieved = counter-measured HBM bytes of ONE training step (FETCH_SIZE x 2 + WRITE_SIZE summed over every kernel "
                               "of the step, separate rocprofv3 --pmc passes of this command) / ms_per_step; the step has no single dominant "
                               "kernel, so the roofline is the step's",
                    inputs=dict(source=src, stale_vs_current_sources=stale, per_kernel=None if pmc is None else pmc.get("per_kernel")),
                    design_bytes_per_step=design_total, design_GBps=design_total / t_step / 1e9, design_breakdown=design,
                    kernels_ms_serialised=kernels, kernels_ms_serialised_sum=sum(kernels.values()),
                    note="kernels_ms_serialised: one step with the side stream off, intervals between events recorded after each library "
                         "call (torch glue in front of a call is charged to it); #k = k-th call of that entry point in the step")
    cpu = None
    if not a.no_cpu_baseline and rk.world == 1:
        try:
            cpu = cpu_baseline_train(cfg, weights, 2048)
        except Exception as e:
            cpu = dict(error=repr(e))
    return dict(metric="rays/sec, training step (forward + backward + FusedAdam + coarse-table refresh)", value=rays_per_s, unit="rays/s",
                samples_per_s=rays_per_s * 384, n_gpus=rk.world, steps=a.steps, warmup=a.warmup, clock_ramp_s=RAMP_SECONDS,
                ms_per_step=t_step * 1e3, rank_step_ms=spread, higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype="f32 (tables, gradients, optimiser state; matrix products as fp16 / bf16 hi+lo MFMA with fp32 accumulate)",
                data="synthetic",
                config=dict(workload="OmniBlender barbershop shape: grid [150,172,516]; 8192 rays x (128 coarse + 128 fine) per step, "
                                     "is_train noise, MSE vs random targets" + (" + TV/L1/ortho/entropy" if a.train_reg else "") +
                                     " (BASELINE configs[3])",
                            rays_per_step_per_gpu=N, parallelism="independent replicas" if rk.world > 1 else "1 GPU"),
                phases_ms=dict(forward=ev[0].elapsed_time(ev[1]), backward=ev[1].elapsed_time(ev[2]), adam_and_refresh=ev[2].elapsed_time(ev[3])),
                step_mode="hipGraph replay of the captured iteration" if graphed is not None else ("eager" if graph_error is None else "eager (graph capture failed: " + graph_error + ")"),
                eager_ms_per_step=dt_eager / a.steps * 1e3,
                loss_first=float(losses[0]), loss_last=float(losses[-1]), peak_mem_GB=torch.cuda.max_memory_allocated() / 2 ** 30,
                roofline=roofline, cpu_baseline=cpu,
                speedup_vs_cpu=None if not cpu or "value" not in cpu else rays_per_s / cpu["value"])


# =====================================================================================================
# --config erp (BASELINE configs[2] and [4])
# =====================================================================================================
ERP_NC, ERP_NF = 128, 128


def erp_pose(k: int, K: int) -> np.ndarray:
    ang = 2 * np.pi * k / max(K, 1)
    c, s = np.cos(ang), np.sin(ang)
    return np.array([[c, 0, s, 0.3 * c], [0, 1, 0, 0.05 * (k % 5)], [-s, 0, c, 0.3 * s]], np.float32)


def cpu_baseline_erp(cfg, weights, H: int, W: int, n_rays: int, alpha_mask=None):
    """The oracle's 128+128 render (resampling, envmap on) of `n_rays` rays of the first view, spread evenly over the image.
    `alpha_mask` = (yin, yang) volumes: applied with TensorBase.forward's semantics (tensorBase.py:464-478), as the HIP path does."""
    from oracle.egonerf_oracle import OracleScene, erp_rays_reference
    logical = os.cpu_count() or 1
    threads = min(32, logical)
    torch.set_num_threads(threads)
    sc = OracleScene(cfg, weights)
    if alpha_mask is not None:
        sc.alpha_mask = alpha_mask
    pose = torch.from_numpy(erp_pose(0, 1))
    allr = erp_rays_reference(H, W, pose)
    pick = torch.linspace(0, allr.shape[0] - 1, n_rays).long()
    rays = allr[pick]
    with torch.no_grad():
        sc.forward(rays[:64], n_coarse=ERP_NC, n_fine=ERP_NF, resampling=True)
        best = float("inf")
        for _ in range(2):
            t = time.perf_counter()
            out = sc.forward(rays, n_coarse=ERP_NC, n_fine=ERP_NF, resampling=True)
            best = min(best, time.perf_counter() - t)
    return dict(value=n_rays / best, unit="rays/s", cores=threads, kind="port",
                sample=f"oracle render of {n_rays} rays (evenly spaced pixels of view 0 of the {H}x{W} image) x ({ERP_NC}+{ERP_NF}) samples, envmap on, "
                       + ("the same alpha mask applied, " if alpha_mask is not None else "") +
                       f"best of 2 ({best:.2f} s) with {threads} ATen threads (survey container, 8 threads, 4096 x (128+128): 1 208 rays/s)"), out, rays, pick


def erp_chunk_split(model, rays_c, reps: int = 12, ERP_NC: int = None, ERP_NF: int = None):
    """The five launches of one ERP chunk (what ego_render_forward queues), event-timed through the stage entry points with the
    model's CURRENT scene (mask / thresholds included) -> ({kernel: ms}, fraction of 32-sample tiles the shade skips)."""
    from egonerf_amd import _lib
    ERP_NC = globals()["ERP_NC"] if ERP_NC is None else ERP_NC
    ERP_NF = globals()["ERP_NF"] if ERP_NF is None else ERP_NF
    dev = rays_c.device
    lib, st = _lib.load(), _lib.stream_handle()
    sc = model.scene()
    N, S = rays_c.shape[0], ERP_NC + ERP_NF
    f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    sched = model._sched(ERP_NC, dev)
    zc, wc, z, w, bg, crd, rgb = f(N, ERP_NC), f(N, ERP_NC), f(N, S), f(N, S), f(N), f(N, S, 4), f(N, S, 3)
    act = torch.empty(N * S // 32 + 1, device=dev, dtype=torch.uint8)
    rgb_map, depth, bgm, envm = f(N, 3), f(N), f(N, 3), f(N, 3)
    near = float(model.near_far[0])
    names = ("k_march_density(coarse)", "k_sample_pdf_merge", "k_march_density(fine)", "k_shade", "k_composite")
    folded = bool(lib.ego_render_forward_folds(sc, N, S))
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(reps)]
    for i in range(reps + 2):
        e = evs[max(i - 2, 0)]
        e[0].record()
        _lib.check(lib.ego_march_density(sc, rays_c.data_ptr(), N, ERP_NC, None, sched.data_ptr(), None, near, 1, zc.data_ptr(), None, 0,
                                         wc.data_ptr(), None, None, None, None, st), "march coarse")
        e[1].record()
        _lib.check(lib.ego_sample_pdf_merge(zc.data_ptr(), wc.data_ptr(), None, N, ERP_NC, ERP_NF, 1, z.data_ptr(), None, st), "pdf merge")
        e[2].record()
        _lib.check(lib.ego_march_density(sc, rays_c.data_ptr(), N, S, z.data_ptr(), None, None, near, 2, None, None, 0, w.data_ptr(), bg.data_ptr(),
                                         crd.data_ptr(), None, act.data_ptr(), st), "march fine")
        e[3].record()
        if folded:   # what ego_render_forward launches here: shading with the compositing in its epilogue ("k_composite" then times nothing)
            _lib.check(lib.ego_shade_composite(sc, rays_c.data_ptr(), z.data_ptr(), crd.data_ptr(), w.data_ptr(), bg.data_ptr(), N, S, act.data_ptr(),
                                               rgb_map.data_ptr(), depth.data_ptr(), bgm.data_ptr(), envm.data_ptr(), st), "shade_composite")
            e[4].record()
        else:
            _lib.check(lib.ego_shade(sc, rays_c.data_ptr(), z.data_ptr(), crd.data_ptr(), N, S, rgb.data_ptr(), None, act.data_ptr(), st), "shade")
            e[4].record()
            _lib.check(lib.ego_composite(sc, rays_c.data_ptr(), z.data_ptr(), w.data_ptr(), bg.data_ptr(), rgb.data_ptr(), N, S, rgb_map.data_ptr(),
                                         depth.data_ptr(), bgm.data_ptr(), envm.data_ptr(), None, st), "composite")
        e[5].record()
    torch.cuda.synchronize()
    ms = np.array([[e[k].elapsed_time(e[k + 1]) for k in range(5)] for e in evs]).mean(0)
    return {n: float(v) for n, v in zip(names, ms)}, float((act[: N * S // 32] == 0).float().mean())


def run_erp(a, rk: Ranks):
    from egonerf_amd.renderer import erp_rays, psnr_from_sse, shard_bounds, volume_renderer
    dev = rk.dev
    H, W = a.erp_size
    over = {} if a.density_shift is None else dict(density_shift=a.density_shift)
    cfg = synth.SceneConfig(**dict(synth.RICOH, **over))
    weights = synth.make_weights(cfg, seed=1234)
    carve = bool(getattr(a, "carve", False))
    if carve:   # real empty space: what the reference's occupancy mask is for (BASELINE configs[2] "empty-space skipping on")
        weights = synth.carve_empty_space(weights, cfg)
    mask_ab = carve and a.mask
    model = synth.build_model(cfg, weights, dev)
    # 16384-ray chunks: the per-chunk workspace (coords, colours, weights: 150 MB at 256 samples) then stays inside the 256 MB Infinity
    # Cache between the march that writes it and the shade / composite that read it (65536: 0.168-0.171 s per image, 16384: 0.165)
    chunk = int(os.environ.get("EGO_ERP_CHUNK", "16384"))
    kw = dict(chunk=chunk, n_coarse=ERP_NC, n_fine=ERP_NF, exp_sampling=True, resampling=True, use_coarse_sample=True, device=dev,
              keep_alpha=False)  # an image render reads rgb only (renderer.py:125-157)
    row0, row1 = shard_bounds(H, rk.world, rk.rank)  # contiguous block of rows per rank
    K = a.views or max(a.steps, 1)
    state = dict(k=0)

    def render(k):
        rays = erp_rays(H, W, erp_pose(k, K), dev, row0, row1 - row0)
        return volume_renderer(rays, model, **kw)[0]

    def step():
        state["last"] = render(state["k"] % K)
        state["k"] += 1

    def all_max(v: float) -> float:
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if rk.shared else dev)
        if rk.dist is not None:
            rk.dist.all_reduce(t, op=rk.dist.ReduceOp.MAX)
        return float(t.item())

    rays_c = erp_rays(H, W, erp_pose(0, K), dev, H // 3, max(1, -(-chunk // W)))[:chunk].contiguous()   # a chunk from the image's middle third
    mask_info = None
    with torch.no_grad():
        occupied = None
        if a.mask:   # the reference's mask construction (EgoNeRF.py:437-489) on this field; applied as TensorBase.forward does (tensorBase.py:464-478)
            occupied = model.updateAlphaMask()
            model.use_alpha_mask = True
        # reference images for the PSNR column: the same views with the fp32-MFMA arithmetic (and the same mask) and no lossy skipping
        default_prec = model.mlp_precision
        model.mlp_precision = "f32"
        refs = [render(k) for k in range(min(K, 2))]
        model.mlp_precision = default_prec
        if mask_ab:   # the same field and views with the mask OFF first (EgoNeRF.forward as written: every sample evaluated)
            model.use_alpha_mask = False
            dt_off = timed(rk, step, a.steps, a.warmup)
            imgs_off = [render(k) for k in range(min(K, 2))]
            split_off, skipped_off = erp_chunk_split(model, rays_c) if rk.rank == 0 else (None, None)
            state["k"] = 0
            model.use_alpha_mask = True
        model.early_termination_eps = a.term_eps
        dt = timed(rk, step, a.steps, a.warmup)
        spread = rk.rank_step_ms(a.steps)
        psnrs = []
        imgs_on = []
        for k, ref in enumerate(refs):
            img = render(k)
            imgs_on.append(img)
            d = img.double() - ref.double()
            stat = torch.stack([(d * d).sum(), torch.tensor(float(d.numel()), device=dev, dtype=torch.float64)])
            if rk.dist is not None:
                stat = stat.cpu() if rk.shared else stat
                rk.dist.all_reduce(stat)
            psnrs.append(psnr_from_sse(max(stat[0].item(), 1e-300), stat[1].item()))
        psnr_same = rk.same_on_all_ranks(psnrs)   # the all-reduced statistics must give every rank the same PSNR (renderer.py:156-157)
        if mask_ab:
            diff = [float((on - off).abs().max()) for on, off in zip(imgs_on, imgs_off)]
            mean = [float((on - off).abs().mean()) for on, off in zip(imgs_on, imgs_off)]
            mask_info = dict(occupied_fraction=occupied, s_per_image_mask_off=dt_off / a.steps, s_per_image_mask_on=dt / a.steps,
                             speedup=dt_off / dt, max_abs_rgb_masked_vs_unmasked=all_max(max(diff)), mean_abs_rgb_masked_vs_unmasked=max(mean),
                             note="mask built by updateAlphaMask (EgoNeRF.py:437-489) from this field's own density, applied with TensorBase.forward's "
                                  "semantics (sigma = 0 where the trilinear mask value is <= 0, tensorBase.py:464-478); the march skips the gather "
                                  "of 64-sample passes that are masked out entirely and the shade skips 32-sample tiles whose weights are all 0; "
                                  "masked vs unmasked differs because 'empty' space still has sigma = softplus(density_shift) in the unmasked render "
                                  "(the reference's rule drops it); parity (below) is against the oracle WITH THE SAME MASK")
        del imgs_on
    if rk.rank != 0:
        return None
    # ---- per-chunk kernel split, event-timed through the stage entry points (the same five launches ego_render_forward queues) ----
    split, tiles_skipped = erp_chunk_split(model, rays_c)
    ms_sum = float(sum(split.values()))
    N, S = rays_c.shape[0], ERP_NC + ERP_NF
    roofline = shade_roofline(model.mlp_precision, split["k_shade"] * 1e-3, N * S)
    pmc, src, stale = load_pmc_section("erp_image")
    t_img = dt / a.steps
    if pmc is not None and pmc.get("traffic_bytes") is not None and not carve:   # whole-image counters replace the headline kernel's per-launch ones
        roofline.update(traffic=pmc["traffic_bytes"], hbm_counter_frac=pmc["traffic_bytes"] / t_img / (HBM_PEAK_GBPS * 1e9),
                        traffic_scope="one whole image (every kernel of every chunk), " + str(src), traffic_stale_vs_current_sources=stale)
    else:
        roofline.update(traffic=None, hbm_counter_frac=None)
    for k in ("issue", "l1", "inputs", "matrix_pipe_busy", "executed_TFLOPs"):   # those are derived from the 4096 x 512 launch's counters
        roofline.pop(k, None)
    roofline.update(chunk_rays=N, chunk_samples_per_ray=S, chunk_kernels_ms=split, chunk_kernels_ms_sum=ms_sum,
                    chunks_per_image=-(-(row1 - row0) * W // chunk), exact_zero_weight_tiles_skipped_frac_in_chunk=tiles_skipped,
                    note="dominant kernel = k_shade on one chunk (flops as in the headline; with tiles skipped `frac` counts the flops of ALL "
                         "samples of the chunk against the time of the tiles that ran, i.e. it is an effective rate); chunk_kernels_ms = the five "
                         "launches of a chunk, event timed through the stage entry points; traffic = counter HBM bytes of one whole image")
    if mask_ab:
        roofline.update(chunk_kernels_ms_mask_off=split_off, exact_zero_weight_tiles_skipped_frac_in_chunk_mask_off=skipped_off)
    cpu = parity = None
    if not a.no_cpu_baseline and rk.world == 1:
        try:
            am = None
            if a.mask and model.alphaMask is not None:
                am = (model.alphaMask.alpha_volume_yin.cpu(), model.alphaMask.alpha_volume_yang.cpu())
            cpu, ref_out, cpu_rays, pick = cpu_baseline_erp(cfg, weights, H, W, 4096, alpha_mask=am)
            with torch.no_grad():
                got = volume_renderer(cpu_rays.to(dev), model, **kw)   # the oracle's own rays: the ray generators are compared in tests/test_hip_ricoh.py
            err = float((got[0].cpu() - ref_out[0]).abs().max())
            mse = float(((got[0].cpu() - ref_out[0]) ** 2).mean())
            dp, p_hip, p_ref = synth.delta_psnr(got[0].cpu().clamp(0, 1).numpy(), ref_out[0].clamp(0, 1).numpy())
            parity = dict(max_abs_rgb_err=err, psnr_vs_oracle_db=float(-10 * np.log10(max(mse, 1e-30))), rays=int(pick.numel()),
                          delta_psnr_db=dp, psnr_hip_vs_gt_db=p_hip, psnr_ref_vs_gt_db=p_ref,
                          alpha_mask_applied_in_both=am is not None,
                          note="a sample within an ulp of a yin/yang border may land on the other grid with another libm (DESIGN.md 2); "
                               "tests/test_hip_ricoh.py handles that case explicitly", tolerance=dict(rgb=1e-4, delta_psnr_db=1e-3))
        except Exception as e:
            cpu = dict(error=repr(e))
    rays_per_s = H * W / t_img
    line = dict(metric="rays/sec, full equirectangular image render (128 coarse + 128 fine samples, envmap on)", value=rays_per_s,
                unit="rays/s", samples_per_s=rays_per_s * 384, n_gpus=rk.world, steps=a.steps, warmup=a.warmup, clock_ramp_s=RAMP_SECONDS,
                ms_per_step=t_img * 1e3, s_per_image=t_img, rank_step_ms=spread, psnr_identical_on_all_ranks=psnr_same,
                row_shards=[list(shard_bounds(H, rk.world, r)) for r in range(rk.world)],
                higher_is_better=True, scaling="strong", vs_baseline=None,
                dtype=f"f32 (matrix products: mlp_precision = {model.mlp_precision}, fp32 accumulate)", data="synthetic",
                config=dict(workload=f"Ricoh360-like scene (near_far [0.1,300], r0 0.05, shift {cfg.density_shift:g}, envmap 3x3840x1920, grid [150,172,516]"
                                     + (", density carved to two radial shells minus a phi wedge: real empty space" if carve else "") + "); "
                                     f"a step = one {H}x{W} ERP image, rays generated on the device, rows sharded over the ranks, exact zero-weight "
                                     f"tile skip on" + (", occupancy-grid empty-space skipping ON" if a.mask else "") +
                                     " (BASELINE configs[2]; configs[4] at --gpus 8)",
                            alpha_mask=bool(a.mask), alpha_mask_occupied_fraction=occupied, term_eps=a.term_eps,
                            parallelism=f"row-sharded x{rk.world}"),
                psnr_vs_f32_unskipped_db=psnrs, roofline=roofline, cpu_baseline=cpu, parity=parity,
                speedup_vs_cpu=None if not cpu or "value" not in cpu else rays_per_s / cpu["value"])
    if mask_info is not None:
        line["mask"] = mask_info
    return line


def run_other_shape(a, rk: Ranks):
    """A model shape the tuned kernels do not serve - `shadingMode="MLP"` (MLPRender, tensorBase.py:100-126) over the shipped tables - through
    the any-shape kernels (csrc/ego_generic.hip: fp32 MFMA since round 6): inference at the headline's 4096 x 512, one training forward +
    backward at configs[3]'s 8192 x (128 + 128), parity of the render against the oracle.  VERDICT r05 weak #8 in the record."""
    dev = rk.dev
    cfg = synth.SceneConfig(shadingMode="MLP")
    weights = synth.make_weights(cfg, seed=1234)
    model = synth.build_model(cfg, weights, dev)
    rays = torch.from_numpy(synth.make_rays(N_RAYS, seed=1)).to(dev)
    with torch.no_grad():
        dt = timed(rk, lambda: model(rays, n_coarse=N_SAMPLES, exp_sampling=True), a.steps, a.warmup)
        got = model(rays[:128], n_coarse=N_SAMPLES, exp_sampling=True)[0].cpu()
    from oracle.egonerf_oracle import OracleScene
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = OracleScene(cfg, weights).forward(rays[:128].cpu(), n_coarse=N_SAMPLES)[0]
    err = float((got - ref).abs().max())
    model.train()
    rays8 = torch.from_numpy(synth.make_rays(TRAIN_RAYS, seed=1)).to(dev)
    gt = torch.rand(TRAIN_RAYS, 3, device=dev)

    def step():
        model.zero_grad(set_to_none=True)
        rgb, *_ = model(rays8, is_train=True, n_coarse=TRAIN_NC, n_fine=TRAIN_NF, exp_sampling=True, resampling=True, use_coarse_sample=True)
        torch.mean((rgb - gt) ** 2).backward()
    dtt = timed(rk, step, 5, 2)
    t = dt / a.steps
    # algorithmic flops per sample: basis 2 x 144 x app_dim + MLPRender (in_c -> 128 -> 128 -> 3)
    in_c = model.head_in_mlpC
    flop = 2.0 * (144 * cfg.app_dim + in_c * 128 + 128 * 128 + 128 * 3)
    achieved = flop * N_RAYS * N_SAMPLES / t / 1e12
    return dict(metric="rays/sec at 4096-ray batch, 512 samples, shadingMode MLP (any-shape kernels)", value=N_RAYS / t, unit="rays/s",
                ms_per_step=t * 1e3, steps=a.steps, warmup=a.warmup, dtype="f32 (fp32-input MFMA)", data="synthetic",
                train_fwd_bwd_ms=dtt / 5 * 1e3, tuned=bool(model.is_tuned_shape),
                config=dict(workload=f"{N_RAYS} x {N_SAMPLES} render and {TRAIN_RAYS} x ({TRAIN_NC}+{TRAIN_NF}) fwd+bwd of a shadingMode='MLP' model on the headline grid"),
                roofline=dict(bound="mfma", kernel="k_shade_generic (whole step timed: march + shade + composite)", unit="TFLOP/s", achieved=achieved,
                              peak=MFMA_F32_PEAK_TFLOPS, frac=achieved / MFMA_F32_PEAK_TFLOPS, traffic=None,
                              note="algorithmic flops of basis + MLPRender per sample over the WHOLE step's time (an underestimate of the kernel's own rate)"),
                parity=dict(max_abs_rgb_err=err, tolerance_rgb=1e-4, rays=128))


def run_secondary(a, rk: Ranks):
    """Short runs of the other single-GPU BASELINE configs, folded into the default line as `secondary` (N = 1 only): configs[3]
    (training step) and configs[2] (full ERP image; once more on an opaque field, density_shift 0, which is what a trained scene
    looks like to the exact zero-weight tile skip).  Each carries its own roofline and cpu_baseline."""
    import copy
    out = {}

    def sub(**over):
        b = copy.copy(a)
        for k, v in over.items():
            setattr(b, k, v)
        return b

    jobs = (("render_config1_256x64", run_render_shape, sub(config="render", steps=200, warmup=10, shape=[256, 64, 0])),
            ("render_resampling_4096x256+256", run_render_shape, sub(config="render", steps=100, warmup=10, shape=[4096, 256, 256])),
            ("render_fresh_rays", run_render_variant, sub(config="render", steps=128, warmup=8, fresh_rays=64, n_voxel=None)),
            ("render_big_grid", run_render_variant, sub(config="render", steps=64, warmup=8, fresh_rays=64, n_voxel=216e6)),
            ("train", run_train, sub(config="train", steps=5, warmup=2, train_reg=False)),
            ("erp", run_erp, sub(config="erp", steps=2, warmup=1, views=2, erp_size=[1024, 2048], mask=False, carve=False, term_eps=0.0, density_shift=None)),
            ("erp_masked", run_erp, sub(config="erp", steps=2, warmup=1, views=2, erp_size=[1024, 2048], mask=True, carve=True, term_eps=0.0,
                                        density_shift=None)),
            ("erp_opaque_field", run_erp, sub(config="erp", steps=2, warmup=1, views=2, erp_size=[1024, 2048], mask=False, term_eps=0.0,
                                              density_shift=0.0, carve=False)),   # with its own parity leg + cpu_baseline
            ("eval_metrics_1024x2048", run_eval_metrics, sub(config="metrics", steps=10, warmup=2, erp_size=[1024, 2048])),
            ("other_shape_mlp_head", run_other_shape, sub(config="render", steps=20, warmup=3)))
    for name, fn, args in jobs:
        t0 = time.perf_counter()
        try:
            torch.cuda.empty_cache()
            line = fn(args, rk)
            line["wall_s_incl_setup_and_cpu_baseline"] = time.perf_counter() - t0
            out[name] = line
        except Exception as e:  # a secondary must never cost the headline its line
            out[name] = dict(error=repr(e))
    return out


# =====================================================================================================
# the line the driver parses: compact (a 27.6 KB line did not survive the driver's 8 KB stdout tail)
# =====================================================================================================
COMPACT_LIMIT = 6000     # bytes; the driver keeps the last 8 KB of stdout
_ROOFLINE_KEYS = ("bound", "kernel", "unit", "achieved", "peak", "frac", "traffic", "ms", "hbm_counter_frac", "matrix_pipe_busy")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _sig(v, digits: int = 6):
    """Floats to `digits` significant digits (the full-precision values are in the full record)."""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}") if np.isfinite(v) else None
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def _clip(text, n: int):
    return text if not isinstance(text, str) or len(text) <= n else text[: n - 1] + "\u2026"


def _physical(rf: dict) -> dict:
    """A roofline object whose `frac` is a physical fraction.  Under `bound: "hbm"` `achieved` / `frac` are COUNTER
    bytes / time / 8 TB/s; a record that still carries SURVEY 8(d)'s algorithmic tap bytes there (cache-served taps: > 1) has that figure
    moved to `algorithmic_frac` and the counter fraction - or null - put in its place.  No roofline in the line may exceed 1."""
    rf = dict(rf)
    f = rf.get("frac")
    if rf.get("bound") == "hbm" and isinstance(f, (int, float)) and f > 1.0:
        rf.setdefault("algorithmic_frac", f)
        c = rf.get("hbm_counter_frac")
        ok = isinstance(c, (int, float)) and np.isfinite(c) and c <= 1.0
        rf["frac"] = c if ok else None
        rf["achieved"] = rf["traffic"] / (rf["ms"] * 1e-3) / 1e9 if ok and rf.get("traffic") and rf.get("ms") else (c * rf["peak"] if ok and rf.get("peak") else None)
    return rf


def _brief_roofline(rf):
    if not isinstance(rf, dict):
        return None
    rf = _physical(rf)
    out = {k: rf[k] for k in _ROOFLINE_KEYS if k in rf}
    out.setdefault("traffic", None)
    if rf.get("algorithmic_frac") is not None:
        out["algorithmic_frac"] = rf["algorithmic_frac"]
    # what binds the kernel, in the driver's record itself: SIMD issue time and vector-L1 path time over kernel time
    for k in ("issue", "l1"):
        if isinstance(rf.get(k), dict) and rf[k].get("frac") is not None:
            out[k + "_frac"] = rf[k]["frac"]
    src = (rf.get("inputs") or {}).get("source") or rf.get("traffic_scope")
    # which fields this process measured and which it read from the tracked counter passes (bench.py cannot collect PMC counters itself)
    out["measured_here"] = ["ms", "achieved", "frac"]
    out["from_counter_pass"] = None if out.get("traffic") is None else dict(
        file=_clip(src, 80), fields=[k for k in ("traffic", "hbm_counter_frac", "matrix_pipe_busy") if out.get(k) is not None],
        stale=bool((rf.get("inputs") or {}).get("stale_vs_current_sources", rf.get("traffic_stale_vs_current_sources", False))))
    return out


def _brief_cpu(cb):
    if not isinstance(cb, dict):
        return None
    out = {k: cb.get(k) for k in _CPU_KEYS}
    out["sample"] = _clip(out.get("sample"), 160)
    return out


def _brief_parity(p):
    if not isinstance(p, dict):
        return None
    return {k: p[k] for k in ("max_abs_rgb_err", "delta_psnr_db", "psnr_hip_vs_gt_db", "psnr_ref_vs_gt_db", "max_abs_depth_err", "abs_ssim_err", "abs_psnr_err_db", "rays", "tolerance") if k in p}


def compact_line(full: dict, full_paths=()) -> dict:
    """The contract's keys + roofline + cpu_baseline + parity of the headline, and value / ms_per_step / roofline.frac / parity of every
    secondary; everything else (notes, instruction counts, alternative arithmetics, per-kernel tables) stays in the full record."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "data",
            "samples_per_s", "s_per_image", "process_group", "speedup_vs_cpu", "loss_first", "loss_last", "psnr_identical_on_all_ranks", "row_shards")
    out = {k: full[k] for k in keep if k in full}
    out["dtype"] = _clip(full.get("dtype"), 120)
    cfg = dict(full.get("config") or {})
    cfg["workload"] = _clip(cfg.get("workload"), 200)
    out["config"] = cfg
    out["roofline"] = _brief_roofline(full.get("roofline"))
    out["cpu_baseline"] = _brief_cpu(full.get("cpu_baseline"))
    out["parity"] = _brief_parity(full.get("parity"))
    if full.get("rank_step_ms"):
        out["rank_step_ms"] = {k: full["rank_step_ms"][k] for k in ("min", "max", "per_rank") if k in full["rank_step_ms"]}
    if "psnr_vs_f32_unskipped_db" in full:
        out["psnr_vs_f32_unskipped_db"] = full["psnr_vs_f32_unskipped_db"][:2]
    sec = full.get("secondary")
    if isinstance(sec, dict):
        brief = {}
        for name, ln in sec.items():
            if "error" in ln:
                brief[name] = dict(error=_clip(ln["error"], 120))
                continue
            rf = _physical(ln.get("roofline") or {})
            b = dict(value=ln.get("value"), unit=ln.get("unit"), ms_per_step=ln.get("ms_per_step"), steps=ln.get("steps"),
                     roofline=dict(bound=rf.get("bound"), frac=rf.get("frac"), traffic=rf.get("traffic"), unit=rf.get("unit"), achieved=rf.get("achieved")))
            if rf.get("algorithmic_frac") is not None:
                b["roofline"]["algorithmic_frac"] = rf["algorithmic_frac"]
            if isinstance(ln.get("cpu_baseline"), dict):
                b["cpu_baseline"] = dict(value=ln["cpu_baseline"].get("value"), cores=ln["cpu_baseline"].get("cores"), kind=ln["cpu_baseline"].get("kind"))
            if isinstance(ln.get("parity"), dict):
                b["parity"] = {k: ln["parity"][k] for k in ("max_abs_rgb_err", "delta_psnr_db", "abs_ssim_err", "abs_psnr_err_db") if k in ln["parity"]}
            brief[name] = b
        out["secondary"] = brief
    out["full_record"] = [os.path.relpath(p, REPO) if p.startswith(REPO) else p for p in full_paths]
    out = _sig(out)
    text = json.dumps(out, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:      # never let the line outgrow the driver's window again: drop the optional blocks, largest first
        for k in ("secondary", "rank_step_ms", "row_shards", "full_record"):
            out.pop(k, None)
            if len(json.dumps(out, separators=(",", ":"))) <= COMPACT_LIMIT:
                break
    return out


def write_full(full: dict, a) -> list:
    paths = [a.full_out] if a.full_out else [os.path.join(REPO, "bench_full.json")]
    scratch = os.path.join(REPO, "gpurun_out")
    if not a.full_out and os.path.isdir(scratch):
        paths.append(os.path.join(scratch, "bench_full.json"))
    done = []
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
            done.append(p)
        except OSError:
            pass
    return done


def main():
    a = parse()
    if a.cpu_worker:
        return cpu_worker(*a.cpu_worker)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)
    rk = Ranks(a)
    variant = a.config == "render" and (a.n_voxel or a.fresh_rays or a.shape)
    line = (run_render_shape if (a.config == "render" and a.shape) else run_render_variant if variant
            else dict(render=run_render, train=run_train, erp=run_erp, metrics=run_eval_metrics)[a.config])(a, rk)
    if a.config == "render" and not variant and rk.world == 1 and not a.no_secondary and a.density_shift is None:
        line["secondary"] = run_secondary(a, rk)
    if rk.rank == 0:
        line["process_group"] = None if rk.dist is None else rk.dist.get_backend()   # "nccl" = RCCL on ROCm
        paths = write_full(line, a)
        print(json.dumps(compact_line(line, paths), separators=(",", ":")), flush=True)
    rk.finish()


if __name__ == "__main__":
    main()
