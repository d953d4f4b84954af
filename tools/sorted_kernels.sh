#!/bin/bash
# Runs ON the GPU box: per-kernel times of tools/sorted_probe.py's scatter calls under rocprofv3 --kernel-trace --stats.
# Usage: tools/sorted_kernels.sh [env assignments ...]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sk
env EGO_SKIP_SELFTEST=1 PROBE_ONLY=1 "$@" rocprofv3 --kernel-trace --stats -d /tmp/sk -o sk -- python $GRAFT_REPO_ROOT/tools/sorted_probe.py > /tmp/sk_out.txt 2>&1
tail -4 /tmp/sk_out.txt
python - <<PY
import sqlite3,glob
p=glob.glob("/tmp/sk/**/*.db", recursive=True)[0]
db=sqlite3.connect(p)
for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 24"):
    name=name.replace("(anonymous namespace)::","").replace("void ","")
    print(f"{name[:80]:80s} {calls:5d} {avg:10.2f} us avg {tot:12.1f} {pct:5.1f}%")
PY
