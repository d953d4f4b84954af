#!/bin/bash
# Same-box A/B of the headline step over several built trees: tools/ab_dirs.sh N dir1 dir2 ...  ("." = the working tree)
N=$1; shift
cd $GRAFT_REPO_ROOT
export EGO_SKIP_SELFTEST=1 EGO_ALLOW_STALE_LIB=1
for i in $(seq $N); do
  for t in "$@"; do
    (cd $GRAFT_REPO_ROOT/$t && timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary --full-out /tmp/ab.json > /dev/null 2>&1
     python -c "
import json; d=json.load(open('/tmp/ab.json')); r=d['roofline']
print('%-10s' % '$t', 'step', round(d['ms_per_step'],4), 'shade', round(r.get('ms',0),4), 'march', round(r['other_kernels_ms']['k_march_density'],4))")
  done
done
