"""Per-entry-point attribution of a `EGO_ROCTX=1 rocprofv3 --marker-trace --kernel-trace` run (SURVEY 5 tracing row).

    python tools/roctx_summary.py <dir-or-.db> [> profiles/rNN/roctx_summary.txt]

rocprofv3 records every kernel dispatch with the stack id of the innermost roctx range open on the launching thread
(`kernels.parent_stack_id` -> `regions.stack_id`); the library opens one range per C-ABI entry point (csrc/ego_host.h, EGO_TRACE), so
the table below is "GPU time per row of SURVEY 8(a)" for whatever the traced command ran - bench.py's headline and every secondary in
one pass, no stage probes."""
import collections
import glob
import json
import os
import sqlite3
import sys


def summarise(db_path: str) -> str:
    c = sqlite3.connect(db_path)
    regions = {}
    for sid, ext, parent in c.execute("select stack_id, extdata, parent_stack_id from regions"):
        try:
            regions[sid] = (json.loads(ext).get("message", "?"), parent)
        except (TypeError, ValueError):
            regions[sid] = ("?", parent)

    def chain(sid):   # innermost first
        out = []
        while sid in regions and len(out) < 8:
            out.append(regions[sid][0])
            sid = regions[sid][1]
        return out

    per_range = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    per_outer = collections.defaultdict(float)
    total = 0.0
    for name, dur, parent in c.execute("select name, duration, parent_stack_id from kernels"):
        ch = chain(parent)
        key = ch[0] if ch else "(outside any ego_* range: torch / runtime kernels)"
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:58]
        per_range[key][0] += 1
        per_range[key][1] += dur
        per_range[key][2][short] += dur
        per_outer[ch[-1] if ch else key] += dur
        total += dur
    lines = [f"== {os.path.basename(db_path)}: {sum(v[0] for v in per_range.values())} kernel dispatches, {total / 1e6:.2f} ms of GPU time, "
             f"{len(regions)} roctx ranges",
             f"{'entry point (innermost roctx range)':40s} {'kernels':>8s} {'GPU ms':>10s} {'share':>7s}   kernels inside (ms)"]
    for key, (n, dur, names) in sorted(per_range.items(), key=lambda kv: -kv[1][1]):
        inside = ", ".join(f"{k} {v / 1e6:.2f}" for k, v in names.most_common(3))
        lines.append(f"{key[:40]:40s} {n:8d} {dur / 1e6:10.3f} {100 * dur / max(total, 1):6.1f}%   {inside}")
    lines.append("")
    lines.append("by OUTERMOST range (ego_render_forward nests the stage entry points):")
    for key, dur in sorted(per_outer.items(), key=lambda kv: -kv[1]):
        lines.append(f"  {key[:60]:60s} {dur / 1e6:10.3f} ms")
    return "\n".join(lines)


if __name__ == "__main__":
    target = sys.argv[1]
    dbs = [target] if target.endswith(".db") else sorted(glob.glob(os.path.join(target, "**", "*.db"), recursive=True))
    if not dbs:
        raise SystemExit(f"no rocprofv3 .db under {target}")
    print("\n\n".join(summarise(p) for p in dbs))
