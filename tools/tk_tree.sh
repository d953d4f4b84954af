#!/bin/bash
# per-kernel training table for a tree: /tmp/tk.sh <dir>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp
(cd $GRAFT_REPO_ROOT/$1 && env EGO_TRAIN_SIDE_STREAM=0 EGO_SKIP_SELFTEST=1 EGO_ALLOW_STALE_LIB=1 rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python bench.py --config train --steps 10 --no-cpu-baseline --full-out /dev/null > /tmp/tp_line.json 2>/dev/null)
python - <<PY
import sqlite3,glob
p=glob.glob("/tmp/tp/**/*.db", recursive=True)[0]
db=sqlite3.connect(p)
for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 8"):
    name=name.replace("(anonymous namespace)::","").replace("void ","")
    if "shade" in name or "wgrad_h" in name: print("$1", f"{name[:60]:60s} {calls:5d} {avg:10.1f} us")
PY
