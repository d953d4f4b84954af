"""gpurun_out/prof_<tag>/pmc_*/**.db (tools/profile_gpu.sh) -> profiles/<round>/pmc_traffic.json: per-launch counter means for the
render kernels, the kernel duration of the GRBM pass (-> effective clock) and the HBM traffic bench.py reports as
roofline.traffic (FETCH_SIZE is in KB and counts 32-byte requests as 64 on gfx950: read bytes = FETCH_SIZE x 1024 x 2,
/opt/skills/guides/MI355X_MICROARCH.md; WRITE_SIZE x 1024).  `_source_hash` records the library sources the counters belong
to (egonerf_amd.build.source_hash of the working tree: run this right after the profile, before editing kernels).

    python tools/pmc_traffic.py <tag> [round] [--out=file]       e.g.  python tools/pmc_traffic.py v1 r02
(tools/profile_gpu.sh runs it on the GPU box with --out=gpurun_out/prof_<tag>/pmc_traffic.json and then deletes the rocprofv3
databases, which exceed what gpurun copies back; copy that file to profiles/<round>/pmc_traffic.json)
"""
import glob, json, os, sqlite3, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from egonerf_amd.build import source_hash
argv = [a for a in sys.argv[1:] if not a.startswith("--out=")]
out_path = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")), None)  # on the GPU box: write next to the passes
tag = argv[0] if len(argv) > 0 else "v1"
rnd = argv[1] if len(argv) > 1 else "r02"
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
names = {"k_shade_h<SHADE>": "%k_shade_h<0, false, false, 0, %", "k_shade_h<SHADE,f16f8>": "%k_shade_h<0, false, false, 1, %", "k_shade_h<SHADE,f16f6>": "%k_shade_h<0, false, false, 2, %",
         "k_shade<SHADE>": "%k_shade<0%", "k_march_density<16>": "%k_march_density%",
         "k_composite": "%k_composite%"}
keys = {"FETCH_SIZE": "FETCH_SIZE_KB", "WRITE_SIZE": "WRITE_SIZE_KB", "TCC_HIT_sum": "TCC_HIT", "TCC_MISS_sum": "TCC_MISS",
        "TCP_TCC_READ_REQ_sum": "TCP_TCC_READ_REQ", "TCP_TOTAL_CACHE_ACCESSES_sum": "TCP_TOTAL_CACHE_ACCESSES", "TA_BUSY_avr": "TA_BUSY_avr",
        "GRBM_GUI_ACTIVE": "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU": "SQ_INSTS_VALU_per_SE", "SQ_INSTS_MFMA": "SQ_INSTS_MFMA_per_SE",
        "SQ_VALU_MFMA_BUSY_CYCLES": "SQ_VALU_MFMA_BUSY_CYCLES_per_SE", "SQ_WAIT_ANY": "SQ_WAIT_ANY_per_SE", "SQ_WAVE_CYCLES": "SQ_WAVE_CYCLES_per_SE",
        "SQ_INSTS_LDS": "SQ_INSTS_LDS_per_SE", "SQ_INSTS_VMEM_RD": "SQ_INSTS_VMEM_RD_per_SE", "SQ_INSTS_SALU": "SQ_INSTS_SALU_per_SE",
        "SQ_BUSY_CYCLES": "SQ_BUSY_CYCLES_per_SE"}
out = {k: {} for k in names}
for p in glob.glob(src + "/pmc_*/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    for short, like in names.items():
        q = "select counter_name, avg(counter_value), avg(duration) from pmc_events where name like ? group by counter_name"
        for ctr, val, dur in db.execute(q, (like,)):
            if ctr in keys:
                out[short][keys[ctr]] = round(val, 1)
            if ctr == "GRBM_GUI_ACTIVE":
                out[short]["duration_us"] = round(dur / 1e3, 3)  # of the same dispatches the GRBM counter was read for
out = {k: v for k, v in out.items() if v}
for short, d in out.items():
    if "FETCH_SIZE_KB" in d and "WRITE_SIZE_KB" in d:
        d["hbm_read_bytes_corrected"] = d["FETCH_SIZE_KB"] * 1024 * 2
        d["hbm_write_bytes"] = d["WRITE_SIZE_KB"] * 1024
        d["traffic_bytes"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]


def whole_step(prefix: str, unit_like: str, what: str):
    """HBM bytes of ONE step of a multi-kernel config: per kernel name, mean bytes per dispatch x dispatches per step, summed.
    Dispatches per step = dispatch rows of the kernel / dispatch rows of `unit_like` (a kernel launched exactly once per step:
    k_adam for the training step, k_erp_rays for an image)."""
    per = {}
    units = None
    for ctr, key, scale in (("FETCH_SIZE", "read_bytes", 1024 * 2), ("WRITE_SIZE", "write_bytes", 1024)):
        dbs = glob.glob(src + f"/{prefix}_{ctr}/**/*.db", recursive=True)
        if not dbs:
            return None
        db = sqlite3.connect(dbs[0])
        rows = list(db.execute("select name, avg(counter_value), count(*) from pmc_events where counter_name = ? group by name", (ctr,)))
        unit_rows = [n for name, _v, n in rows if unit_like in name]
        if not unit_rows:
            return None
        units = unit_rows[0]
        for name, val, n in rows:
            d = per.setdefault(name, {"read_bytes": 0.0, "write_bytes": 0.0, "dispatches_per_step": n / units})
            d[key] = val * scale * n / units
    total_r = sum(d["read_bytes"] for d in per.values())
    total_w = sum(d["write_bytes"] for d in per.values())
    top = sorted(per.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]))[:14]
    return {"hbm_read_bytes_corrected": total_r, "hbm_write_bytes": total_w, "traffic_bytes": total_r + total_w, "steps_profiled": units,
            "what": what,
            "per_kernel": {k[:80]: {kk: round(vv, 1) for kk, vv in v.items()} for k, v in top}}


for key, prefix, unit_like, what in (("train_step", "pmctrain", "k_adam", "one training step of bench.py --config train (8192 rays x (128+128), fwd + bwd + FusedAdam + coarse refresh)"),
                                     ("erp_image", "pmcerp", "k_erp_rays", "one 1024 x 2048 image of bench.py --config erp (128+128 samples, envmap, 16384-ray chunks)")):
    sec = whole_step(prefix, unit_like, what)
    if sec is not None:
        out[key] = sec
# render variants (bench.py --fresh-rays / --n-voxel): HBM bytes of the two grid-sample kernels per step
for key, prefix, what in (("big_grid", "pmcbig", "one step of bench.py --n-voxel 216e6 --fresh-rays 64 (grid [300,346,1036], ~400 MB of tables): k_march_density + k_shade_h"),
                          ("fresh_rays", "pmcfresh", "one step of bench.py --fresh-rays 64 (headline grid, a different ray batch every step): k_march_density + k_shade_h")):
    # one shade launch per step (k_composite is folded into it since round 5 wherever whole rays divide evenly: it no longer marks a step)
    sec = whole_step(prefix, "k_shade_h", what)
    if sec is not None:
        pk = {k: v for k, v in sec["per_kernel"].items() if "k_march_density" in k or "k_shade" in k}
        sec["per_kernel"] = pk
        sec["hbm_read_bytes_corrected"] = sum(v["read_bytes"] for v in pk.values())
        sec["hbm_write_bytes"] = sum(v["write_bytes"] for v in pk.values())
        sec["traffic_bytes"] = sec["hbm_read_bytes_corrected"] + sec["hbm_write_bytes"]
        out[key] = sec
out["_source"] = f"gpurun_out/prof_{tag} (tools/profile_r03.sh / profile_r04.sh {tag}; separate rocprofv3 --pmc passes of `bench.py --steps 10 --warmup 2`, `--config train`, `--config erp`)"
out["_source_hash"] = source_hash()
dst = out_path or os.path.join(root, "profiles", rnd, "pmc_traffic.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v.get("traffic_bytes") for k, v in out.items() if isinstance(v, dict)}))
