#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE passes of bench.py --config train only -> gpurun_out/prof_<tag>/pmc_traffic.json (train_step section)
TAG=${1:-t1}; RND=${2:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp EGO_SKIP_SELFTEST=1 EGO_BENCH_RAMP_SECONDS=0.05
B="python $ROOT/bench.py --no-cpu-baseline --no-secondary"
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d "$OUT/pmctrain_$pmc" -o pmc -- $B --config train --steps 4 --warmup 1 > "$OUT/pmctrain_$pmc.log" 2>&1 || echo "train pmc pass failed: $pmc"
done
python "$ROOT/tools/pmc_traffic.py" "$TAG" "$RND" --out="$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.log" 2>&1
find "$OUT" -name "*.db" -delete
python - <<PY
import json
d=json.load(open("$OUT/pmc_traffic.json"))["train_step"]
print("train step traffic GB:", round(d["traffic_bytes"]/1e9,2), "read", round(d["hbm_read_bytes_corrected"]/1e9,2), "write", round(d["hbm_write_bytes"]/1e9,2))
for k,v in sorted(d["per_kernel"].items(), key=lambda kv:-(kv[1]["read_bytes"]+kv[1]["write_bytes"]))[:16]:
    print(f"  {k[:70]:70s} read {v['read_bytes']/1e9:6.2f} write {v['write_bytes']/1e9:6.2f} x{v['dispatches_per_step']}")
PY
