#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py's step.
# Usage: tools/profile_gpu.sh <tag>     -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-v1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
# the stats pass runs bench.py at its default step count, so that the per-kernel averages are the steady-state ones bench.py reports
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $ROOT/bench.py --no-cpu-baseline > "$OUT/bench_under_trace.log" 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr"; do
  name=$(echo $pmc | tr ' ' '+' | cut -c1-60)
  rocprofv3 --kernel-trace --pmc $pmc -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "pmc pass failed: $pmc" >> "$OUT/errors.log"
done
# compact summaries
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
lines = []
for p in sorted(glob.glob(out + "/trace/**/*.db", recursive=True)):
    db = sqlite3.connect(p)
    lines.append(f"== rocprofv3 --kernel-trace --stats : {os.path.relpath(p, out)}  (top_kernels view; durations in us)")
    lines.append(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        lines.append(f"{name[:70]:70s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")
    lines.append("")
    q = "select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x from kernels where name like '%k_%' group by name"
    for r in db.execute(q):
        lines.append("dispatch resources: " + str(r)[:200])
    lines.append("")
for p in sorted(glob.glob(out + "/pmc_*/**/*.db", recursive=True)):
    db = sqlite3.connect(p)
    lines.append(f"== rocprofv3 --pmc : {os.path.relpath(p, out)}  (mean counter_value per sample row; n = rows; SQ_* rows are per shader engine)")
    q = ("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%k_shade%' or "
         "name like '%k_march%' or name like '%k_composite%' group by name, counter_name")
    for name, ctr, val, n in db.execute(q):
        lines.append(f"{name[:48]:48s} {ctr:32s} {val:18.1f}  n={n}")
    lines.append("")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines)[:7000])
PY
# counters -> compact json next to the passes, then drop the databases (tens of MB; gpurun copies back at most 64 MB)
python "$ROOT/tools/pmc_traffic.py" "$TAG" r02 --out="$OUT/pmc_traffic.json" > /dev/null
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*.csv" -size +1M -delete
