#!/bin/bash
# On the GPU box: the timeline of ONE replayed training step (bench.py --config train, hipGraph): per kernel start / end relative to
# the step's first kernel, stream, and the idle gaps - where the 4.4 ms go when the per-kernel times add up to more than that.
cd /tmp && export TMPDIR=/tmp EGO_SKIP_SELFTEST=1
rm -rf /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --config train --steps 12 --warmup 3 --no-cpu-baseline --full-out /dev/null > /tmp/tl_line.json 2>/dev/null
python - <<'PY'
import sqlite3, glob, json
p = glob.glob("/tmp/tl/**/*.db", recursive=True)[0]
db = sqlite3.connect(p)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in db.execute(f"pragma table_info('{kt}')")]
print("table", kt, cols)
# join with kernel symbol names
sym = [t for t in tabs if "kernel_symbol" in t]
q = f"select start, end, kernel_id, queue_id, stream_id from '{kt}' order by start" if "stream_id" in cols else f"select start, end, kernel_id, queue_id, 0 from '{kt}' order by start"
rows = list(db.execute(q))
names = {}
if sym:
    sc = [r[1] for r in db.execute(f"pragma table_info('{sym[0]}')")]
    nm = "display_name" if "display_name" in sc else "kernel_name"
    for kid, n in db.execute(f"select id, {nm} from '{sym[0]}'"):
        names[kid] = n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
# one step = from the kernel after an k_adam to the next k_adam (inclusive); take the second-to-last complete step
adam = [i for i, r in enumerate(rows) if "k_adam" in names.get(r[2], "")]
a, b = adam[-3] + 1, adam[-2]
t0 = rows[a][0]
busy_end = t0
print(f"step: {b - a + 1} kernels, {(rows[b][1] - t0) / 1e6:.3f} ms from first start to k_adam end")
for s, e, kid, qid, sid in rows[a:b + 1]:
    gap = (s - busy_end) / 1e3
    busy_end = max(busy_end, e)
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{qid % 1000:<4} {'gap %.1f us  ' % gap if gap > 2 else ''}{names.get(kid, kid)}")
PY
