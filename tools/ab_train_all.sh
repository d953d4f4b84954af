#!/bin/bash
# Same-box A/B of the training step over several built trees, every serialised per-call time: tools/ab_train_all.sh N dir1 dir2 ...
N=$1; shift
cd $GRAFT_REPO_ROOT
export EGO_SKIP_SELFTEST=1 EGO_ALLOW_STALE_LIB=1
for i in $(seq $N); do
  for t in "$@"; do
    (cd $GRAFT_REPO_ROOT/$t && timeout 200 python bench.py --config train --steps 30 --no-cpu-baseline --full-out /tmp/abt.json > /dev/null 2>&1
     python -c "
import json; d=json.load(open('/tmp/abt.json')); k=d['roofline']['kernels_ms_serialised']
print('%-12s' % '$t', 'graph', round(d['ms_per_step'],3), 'eager', round(d['eager_ms_per_step'],3), {n.replace('ego_',''): round(v,3) for n,v in k.items() if v > 0.1})")
  done
done
