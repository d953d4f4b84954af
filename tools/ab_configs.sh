#!/bin/bash
# Runs ON the GPU box: alternates pre-built egonerf_amd/libvariant_<name>.so builds under bench.py configs inside one session.
#   tools/ab_configs.sh "xcd0 xcd1" "erp render" [reps]
cd "${GRAFT_REPO_ROOT:-.}"
NAMES=$1; CFGS=$2; REPS=${3:-2}
cp egonerf_amd/libegonerf_hip.so /tmp/shipped.so
for rep in $(seq $REPS); do
  for v in $NAMES; do
    cp egonerf_amd/libvariant_$v.so egonerf_amd/libegonerf_hip.so
    for cfg in $CFGS; do
      python bench.py --config $cfg --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print('$v', '$cfg', 'ms/step', round(d['ms_per_step'], 4), 'value', round(d['value']), r.get('chunk_kernels_ms') or {k: round(v, 4) for k, v in (r.get('other_kernels_ms') or {}).items()}, 'shade', r.get('ms'))"
    done
  done
done
cp /tmp/shipped.so egonerf_amd/libegonerf_hip.so
