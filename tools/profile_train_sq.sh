#!/bin/bash
# Runs on the GPU box: SQ issue / wait counters per kernel of the training step (one stream, eager loop), two separate --pmc passes.
#   tools/profile_train_sq.sh <tag>  -> gpurun_out/trainsq_<tag>.txt
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trainsq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export EGO_BENCH_RAMP_SECONDS=0.05 EGO_TRAIN_SIDE_STREAM=0
B="python $ROOT/bench.py --config train --train-eager --no-cpu-baseline --no-secondary --steps 3 --warmup 1"
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pmc | tr ' ' '+' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pmc -d "$OUT/pmc_$name" -o pmc -- $B > "$OUT/pmc_$name.log" 2>&1 || echo "pass failed: $pmc" >> "$OUT/errors.log"
done
python - "$OUT" <<'PY'
import glob, sqlite3, sys, collections
out = sys.argv[1]
vals = collections.defaultdict(dict)
for p in glob.glob(out + "/pmc_*/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    for name, ctr, val, n in db.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name"):
        vals[name][ctr] = val
lines = ["kernel | waves/SE | issuing | issue-stalled | parked on s_waitcnt | VALU / wave | MFMA / wave | LDS insts / wave | LDS bank-conflict cycles / LDS active"]
for name, c in sorted(vals.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES")
    if not wc or ("k_" not in name):
        continue
    w = max(c.get("SQ_WAVES", 1), 1)
    f = lambda k: c.get(k, float("nan"))
    lines.append(f"{name[:60]:60s} | {w:8.0f} | {f('SQ_ACTIVE_INST_ANY') / wc:.2f} | {f('SQ_WAIT_INST_ANY') / wc:.2f} | {f('SQ_WAIT_ANY') / wc:.2f} | "
                 f"{f('SQ_INSTS_VALU') / w:8.0f} | {f('SQ_INSTS_MFMA') / w:7.0f} | {f('SQ_INSTS_LDS') / w:7.0f} | "
                 f"{f('SQ_LDS_BANK_CONFLICT') / max(f('SQ_LDS_IDX_ACTIVE'), 1):.3f}")
open(out + ".txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find "$OUT" -name "*.db" -delete
