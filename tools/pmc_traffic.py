"""gpurun_out/prof_<tag>/pmc_*/**.db (tools/profile_gpu.sh) -> profiles/r01/pmc_traffic.json: per-launch counter means for the render
kernels and the HBM traffic bench.py reports as roofline.traffic (FETCH_SIZE is in KB and counts 32-byte requests as 64 on gfx950:
read bytes = FETCH_SIZE x 1024 x 2, /opt/skills/guides/MI355X_MICROARCH.md; WRITE_SIZE x 1024)."""
import glob, json, os, sqlite3, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "v9"
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
names = {"k_shade_h<SHADE>": "%k_shade_h<0, false, false>%", "k_march_density<16>": "%k_march_density%", "k_composite": "%k_composite%"}
keys = {"FETCH_SIZE": "FETCH_SIZE_KB", "WRITE_SIZE": "WRITE_SIZE_KB", "TCC_HIT_sum": "TCC_HIT", "TCC_MISS_sum": "TCC_MISS",
        "TCP_TCC_READ_REQ_sum": "TCP_TCC_READ_REQ", "TCP_TOTAL_CACHE_ACCESSES_sum": "TCP_TOTAL_CACHE_ACCESSES", "TA_BUSY_avr": "TA_BUSY_avr",
        "GRBM_GUI_ACTIVE": "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU": "SQ_INSTS_VALU_per_SE", "SQ_INSTS_MFMA": "SQ_INSTS_MFMA_per_SE",
        "SQ_VALU_MFMA_BUSY_CYCLES": "SQ_VALU_MFMA_BUSY_CYCLES_per_SE", "SQ_WAIT_ANY": "SQ_WAIT_ANY_per_SE", "SQ_WAVE_CYCLES": "SQ_WAVE_CYCLES_per_SE"}
out = {k: {} for k in names}
for p in glob.glob(src + "/pmc_*/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    for short, like in names.items():
        for ctr, val in db.execute("select counter_name, avg(counter_value) from pmc_events where name like ? group by counter_name", (like,)):
            if ctr in keys:
                out[short][keys[ctr]] = round(val, 1)
for short, d in out.items():
    if "FETCH_SIZE_KB" in d and "WRITE_SIZE_KB" in d:
        d["hbm_read_bytes_corrected"] = d["FETCH_SIZE_KB"] * 1024 * 2
        d["hbm_write_bytes"] = d["WRITE_SIZE_KB"] * 1024
        d["traffic_bytes"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
out["_source"] = f"gpurun_out/prof_{tag} (tools/profile_gpu.sh {tag}; separate rocprofv3 --pmc passes), build = commit of profiles/r01/{tag}_rocprofv3_summary.txt"
json.dump(out, open(os.path.join(root, "profiles", "r01", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v.get("traffic_bytes") for k, v in out.items() if isinstance(v, dict)}))
