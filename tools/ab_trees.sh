#!/bin/bash
# Runs ON the GPU box: `bench.py --config <cfg>` alternated between this tree and a second checkout under ./_old (built there).
#   tools/ab_trees.sh train 3
cd "${GRAFT_REPO_ROOT:-.}"
CFG=${1:-train}; REPS=${2:-2}
show() { python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = (d.get('roofline') or {}).get('kernels_ms_serialised') or {}
print('$1', 'ms/step', round(d['ms_per_step'], 4), 'eager', round(d.get('eager_ms_per_step') or 0, 4), {n: round(v, 3) for n, v in k.items() if 'shade' in n or 'weight_grad' in n})"; }
for rep in $(seq $REPS); do
  (cd _old && python bench.py --config $CFG --no-cpu-baseline --no-secondary 2>/dev/null | show old)
  python bench.py --config $CFG --no-cpu-baseline --no-secondary 2>/dev/null | show new
done
