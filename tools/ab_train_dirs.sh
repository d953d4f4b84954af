#!/bin/bash
# Same-box A/B of the TRAINING step over several built trees: tools/ab_train_dirs.sh N dir1 dir2 ...  ("." = the working tree)
# prints the graph-replay and eager step times and the serialised per-call times of the scatter entries
N=$1; shift
cd $GRAFT_REPO_ROOT
export EGO_SKIP_SELFTEST=1 EGO_ALLOW_STALE_LIB=1
for i in $(seq $N); do
  for t in "$@"; do
    (cd $GRAFT_REPO_ROOT/$t && timeout 200 python bench.py --config train --steps 40 --no-cpu-baseline --full-out /tmp/abt.json > /dev/null 2>&1
     python -c "
import json; d=json.load(open('/tmp/abt.json')); k=d['roofline']['kernels_ms_serialised']
print('%-10s' % '$t', 'graph', round(d['ms_per_step'],3), 'eager', round(d['eager_ms_per_step'],3), {n: round(v,3) for n,v in k.items() if 'scatter' in n or 'sort' in n})")
  done
done
