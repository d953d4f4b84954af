#!/bin/bash
# Runs ON the GPU box: isolated per-kernel times of the training step (side stream off), rocprofv3 --kernel-trace --stats.
# Usage: tools/train_kernels.sh [extra env assignments ...]   e.g. tools/train_kernels.sh EGO_SCATTER=atomic
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp
env EGO_TRAIN_SIDE_STREAM=0 EGO_SKIP_SELFTEST=1 "$@" rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python $GRAFT_REPO_ROOT/bench.py --config train --steps 10 --no-cpu-baseline --full-out /dev/null > /tmp/tp_line.json 2>/dev/null
python - <<PY
import sqlite3,glob,json
try:
    d=json.loads(open("/tmp/tp_line.json").read().strip().splitlines()[-1]); print("ms_per_step under trace (eager, one stream):", d.get("ms_per_step"))
except Exception as e: print("no line", e)
p=glob.glob("/tmp/tp/**/*.db", recursive=True)[0]
db=sqlite3.connect(p)
for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 22"):
    name=name.replace("(anonymous namespace)::","").replace("void ","")
    print(f"{name[:92]:92s} {calls:5d} {avg:10.1f} us {pct:5.1f}%")
PY
