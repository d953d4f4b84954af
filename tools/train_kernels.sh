#!/bin/bash
# Runs ON the GPU box: isolated per-kernel times of the training step (side stream off), rocprofv3 --kernel-trace --stats.
cd /tmp && export TMPDIR=/tmp
EGO_TRAIN_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python $GRAFT_REPO_ROOT/bench.py --config train --steps 10 > /dev/null 2>&1
python - <<PY
import sqlite3,glob
p=glob.glob("/tmp/tp/**/*.db", recursive=True)[0]
db=sqlite3.connect(p)
for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 10"):
    print(f"{name[:84]:84s} {calls:5d} {avg:12.1f} us {pct:5.1f}%")
PY
