"""Prints the headline and secondary figures of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("headline", d.get("metric", "")[:50], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "shade_ms", r.get("ms"),
      "frac", r.get("frac"), "other", r.get("other_kernels_ms"))
if d.get("parity"):
    print("  parity", d["parity"].get("max_abs_rgb_err"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
for k, v in (d.get("secondary") or {}).items():
    if "error" in v:
        print(" ", k, "ERROR", v["error"][:300]); continue
    rr = v.get("roofline") or {}
    print(" ", k, "value", round(v["value"]), "ms/step", round(v["ms_per_step"], 4), "frac", rr.get("frac"),
          "parity", (v.get("parity") or {}).get("max_abs_rgb_err"), {kk: v[kk] for kk in ("mask", "fresh", "extra") if kk in v})
