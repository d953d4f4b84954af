#!/bin/bash
# On the GPU box: per-kernel time of one training step of a shape that takes the any-shape kernels (tools/generic_profile.py) -> stdout
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp; EGO_TRAIN_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/gp -o gp -- python $ROOT/tools/generic_profile.py > /tmp/gp.log 2>&1 || tail -5 /tmp/gp.log
python - <<'PY'
import sqlite3, glob
p = glob.glob("/tmp/gp/**/*.db", recursive=True)[0]
db = sqlite3.connect(p)
for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 16"):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:92]:92s} {calls:5d} {avg:10.1f} us {pct:5.1f}%")
PY
