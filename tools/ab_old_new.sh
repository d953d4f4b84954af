#!/bin/bash
# Same-box A/B of the headline step: the tree in _ab_old/ (git archive of an earlier commit, built in place) against the working tree,
# alternated N times.  Usage (on the GPU box): tools/ab_old_new.sh [N] [bench args...]
N=${1:-3}; shift
cd $GRAFT_REPO_ROOT
export EGO_SKIP_SELFTEST=1
for i in $(seq $N); do
  for t in old new; do
    d=$GRAFT_REPO_ROOT; [ $t = old ] && d=$GRAFT_REPO_ROOT/_ab_old
    (cd $d && timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary --full-out /tmp/ab_$t.json "$@" > /dev/null 2>&1
     python -c "
import json; d=json.load(open('/tmp/ab_$t.json')); r=d['roofline']
print('$t', 'step', round(d['ms_per_step'],4), 'shade', round(r.get('ms',0),4), 'frac', round(r.get('frac',0),4), r.get('other_kernels_ms') or r.get('kernels_ms'))")
  done
done
