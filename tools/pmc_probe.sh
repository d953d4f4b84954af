#!/bin/bash
# On the GPU box: separate rocprofv3 --pmc passes of a probe script, per-kernel mean counter values.
# Usage: tools/pmc_probe.sh "<python script and args>" <kernel name filter> COUNTERSET [COUNTERSET ...]   (a set = "A B C")
cd /tmp && export TMPDIR=/tmp EGO_SKIP_SELFTEST=1
CMD="$1"; FILT="$2"; shift; shift
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/$CMD > /tmp/pmc_$i.log 2>&1 || echo "pass $i ($set) failed rc=$?"
  python - "$i" "$FILT" <<'PY'
import sqlite3, glob, sys, collections
i, filt = sys.argv[1], sys.argv[2]
p = glob.glob(f"/tmp/pmc_{i}/**/*.db", recursive=True)
if not p: print("no db for pass", i); sys.exit()
db = sqlite3.connect(p[0])
rows = db.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name").fetchall()
for name, ctr, val, n in sorted(rows):
    if filt in name:
        print(f"{name.replace('(anonymous namespace)::','').replace('void ','')[:60]:60s} {ctr:28s} {val:18.1f} n={n}")
PY
done
