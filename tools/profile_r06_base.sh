#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for all three bench configs.
# Usage: tools/profile_r06_base.sh <tag> [round]    -> gpurun_out/prof_<tag>/{summary.txt,pmc_traffic.json,bench_*.log}
#   render : the default `python bench.py` step (4096 x 512), 7 counter passes (HBM, L2, SQ issue / wait, LDS, GRBM, TCP / TA)
#   train  : `bench.py --config train` (8192 x (128+128) fwd + bwd + Adam), FETCH_SIZE and WRITE_SIZE passes -> bytes per STEP
#   erp    : `bench.py --config erp` (1024 x 2048 image, 128+128), FETCH_SIZE and WRITE_SIZE passes -> bytes per IMAGE
# --pmc passes carry --kernel-trace only (no other trace domain), one counter group per run.
set -u
TAG=${1:-v1}
RND=${2:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export EGO_SKIP_SELFTEST=1   # the self-test launches the SHIPPED shade kernels on a tiny scene: it would dilute their per-dispatch averages
B="python $ROOT/bench.py --no-cpu-baseline --no-secondary"
# kernel-trace stats: render at its default step count (steady-state per-kernel averages), train, erp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_render" -o trace -- $B > "$OUT/bench_render_under_trace.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_train" -o trace -- $B --config train --steps 10 --warmup 2 > "$OUT/bench_train_under_trace.log" 2>&1
EGO_TRAIN_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_train_serial" -o trace -- $B --config train --steps 10 --warmup 2 > "$OUT/bench_train_serial_under_trace.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_erp" -o trace -- $B --config erp --steps 2 --warmup 1 > "$OUT/bench_erp_under_trace.log" 2>&1
export EGO_BENCH_RAMP_SECONDS=0.05
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr"; do
  name=$(echo $pmc | tr ' ' '+' | cut -c1-60)
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d "$OUT/pmc_$name" -o pmc -- $B --steps 10 --warmup 2 > "$OUT/pmc_$name.log" 2>&1 || echo "pmc pass failed: $pmc" >> "$OUT/errors.log"
done
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d "$OUT/pmctrain_$pmc" -o pmc -- $B --config train --steps 4 --warmup 1 > "$OUT/pmctrain_$pmc.log" 2>&1 || echo "train pmc pass failed: $pmc" >> "$OUT/errors.log"
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d "$OUT/pmcerp_$pmc" -o pmc -- $B --config erp --steps 1 --warmup 0 --views 1 > "$OUT/pmcerp_$pmc.log" 2>&1 || echo "erp pmc pass failed: $pmc" >> "$OUT/errors.log"
done
# compact summaries
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
lines = []
for p in sorted(glob.glob(out + "/trace_*/**/*.db", recursive=True)):
    db = sqlite3.connect(p)
    lines.append(f"== rocprofv3 --kernel-trace --stats : {os.path.relpath(p, out)}  (top_kernels view; durations in us)")
    lines.append(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if pct >= 0.02 or "k_" in name:
            lines.append(f"{name[:70]:70s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")
    lines.append("")
    q = "select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x from kernels where name like '%k_%' and name not like '%at::native%' group by name"
    for r in db.execute(q):
        lines.append("dispatch resources (name, vgpr, agpr, sgpr, lds, scratch, grid_x, wg_x): " + str(r)[:220])
    lines.append("")
for p in sorted(glob.glob(out + "/pmc_*/**/*.db", recursive=True)):
    db = sqlite3.connect(p)
    lines.append(f"== rocprofv3 --pmc : {os.path.relpath(p, out)}  (mean counter_value per sample row; n = rows; SQ_* rows are per shader engine)")
    q = ("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%k_shade%' or "
         "name like '%k_march%' or name like '%k_composite%' group by name, counter_name")
    for name, ctr, val, n in db.execute(q):
        lines.append(f"{name[:48]:48s} {ctr:32s} {val:18.1f}  n={n}")
    lines.append("")
for p in sorted(glob.glob(out + "/pmctrain_*/**/*.db", recursive=True)) + sorted(glob.glob(out + "/pmcerp_*/**/*.db", recursive=True)):
    db = sqlite3.connect(p)
    lines.append(f"== rocprofv3 --pmc : {os.path.relpath(p, out)}  (per kernel: mean counter_value per dispatch row, rows)")
    for name, ctr, val, n in db.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name order by 3 * 4 desc limit 24"):
        lines.append(f"{name[:64]:64s} {ctr:14s} {val:16.1f}  n={n}")
    lines.append("")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines)[:6000])
PY
# counters -> compact json next to the passes, then drop the databases (tens of MB; gpurun copies back at most 64 MB)
python "$ROOT/tools/pmc_traffic.py" "$TAG" "$RND" --out="$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.log" 2>&1
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*.csv" -size +1M -delete
