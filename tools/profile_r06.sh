#!/bin/bash
# Round 4 additions to tools/profile_r06_base.sh (which it runs first unless SKIP_R03=1): the render variants and the stand-alone
# grid-sample probes.  Usage (on the GPU box): tools/profile_r04.sh <tag>
#   trace_fresh / trace_big   : rocprofv3 --kernel-trace --stats of bench.py --fresh-rays 64 [--n-voxel 216e6]
#   pmcfresh_* / pmcbig_*     : FETCH_SIZE and WRITE_SIZE passes of the same commands -> HBM bytes of march + shade per step
#   trace_probe / trace_probe_big : tools/stage_probe.py (ego_app_feature / ego_density_feature alone) on both grids
set -u
TAG=${1:-v1}
RND=${2:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export EGO_SKIP_SELFTEST=1   # the self-test launches the SHIPPED shade kernels on a tiny scene: it would dilute their per-dispatch averages
B="python $ROOT/bench.py --no-cpu-baseline --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_fresh" -o trace -- $B --fresh-rays 64 --steps 128 --warmup 8 > "$OUT/bench_fresh_under_trace.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_big" -o trace -- $B --fresh-rays 64 --n-voxel 216e6 --steps 64 --warmup 8 > "$OUT/bench_big_under_trace.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_probe" -o trace -- python $ROOT/tools/stage_probe.py > "$OUT/stage_probe.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_probe_big" -o trace -- python $ROOT/tools/stage_probe.py 216e6 > "$OUT/stage_probe_big.log" 2>&1
export EGO_BENCH_RAMP_SECONDS=0.05
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d "$OUT/pmcfresh_$pmc" -o pmc -- $B --fresh-rays 64 --steps 64 --warmup 2 > "$OUT/pmcfresh_$pmc.log" 2>&1 || echo "fresh pmc pass failed: $pmc" >> "$OUT/errors.log"
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d "$OUT/pmcbig_$pmc" -o pmc -- $B --fresh-rays 64 --n-voxel 216e6 --steps 64 --warmup 2 > "$OUT/pmcbig_$pmc.log" 2>&1 || echo "big pmc pass failed: $pmc" >> "$OUT/errors.log"
done
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
lines = []
for p in sorted(glob.glob(out + "/trace_fresh/**/*.db", recursive=True) + glob.glob(out + "/trace_big/**/*.db", recursive=True) + glob.glob(out + "/trace_probe*/**/*.db", recursive=True)):
    db = sqlite3.connect(p)
    lines.append(f"== rocprofv3 --kernel-trace --stats : {os.path.relpath(p, out)}  (top_kernels view; durations in us)")
    lines.append(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if "k_" in name and "at::native" not in name:
            lines.append(f"{name[:70]:70s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")
    lines.append("")
for p in sorted(glob.glob(out + "/pmcfresh_*/**/*.db", recursive=True)) + sorted(glob.glob(out + "/pmcbig_*/**/*.db", recursive=True)):
    db = sqlite3.connect(p)
    lines.append(f"== rocprofv3 --pmc : {os.path.relpath(p, out)}  (per kernel: mean counter_value per dispatch row, rows)")
    for name, ctr, val, n in db.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%k_%' group by name, counter_name order by 3 * 4 desc limit 12"):
        lines.append(f"{name[:64]:64s} {ctr:14s} {val:16.1f}  n={n}")
    lines.append("")
for f in ("stage_probe.log", "stage_probe_big.log"):
    try:
        lines.append(f"== {f}: " + [l for l in open(os.path.join(out, f)).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        lines.append(f"== {f}: missing ({e})")
open(os.path.join(out, "summary_variants.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines)[:5000])
PY
if [ "${SKIP_R03:-0}" != "1" ]; then
  bash "$ROOT/tools/profile_r06_base.sh" "$TAG" "$RND"      # render / train / erp passes + pmc_traffic.json (reads the pmcfresh_ / pmcbig_ passes above too)
else
  python "$ROOT/tools/pmc_traffic.py" "$TAG" "$RND" --out="$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.log" 2>&1
  find "$OUT" -name "*.db" -delete
fi
