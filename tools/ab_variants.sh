#!/bin/bash
# Runs ON the GPU box (via gpurun): alternates the pre-built egonerf_amd/libvariant_<name>.so builds under bench.py inside one session.
#   tools/ab_variants.sh "base all" [reps] [pytest-target]
cd "${GRAFT_REPO_ROOT:-.}"
NAMES=$1; REPS=${2:-3}; TESTS=${3:-}
for v in $NAMES; do
  cp egonerf_amd/libvariant_$v.so egonerf_amd/libegonerf_hip.so
  if [ -n "$TESTS" ]; then echo "== tests $v"; timeout 900 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -3; fi
done
for rep in $(seq $REPS); do
  for v in $NAMES; do
    cp egonerf_amd/libvariant_$v.so egonerf_amd/libegonerf_hip.so
    echo -n "$v: "; python tools/bench_brief.py 1
  done
done
