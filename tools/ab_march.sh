#!/bin/bash
# Runs ON the GPU box: tools/march_timing.py (three shapes + output checksums) alternated over pre-built libvariant_<name>.so builds.
#   tools/ab_march.sh "base rc1" [reps]
cd "${GRAFT_REPO_ROOT:-.}"
cp egonerf_amd/libegonerf_hip.so /tmp/shipped.so
for rep in $(seq ${2:-2}); do
  for v in $1; do cp egonerf_amd/libvariant_$v.so egonerf_amd/libegonerf_hip.so; echo -n "$v: "; python tools/march_timing.py 2>&1 | tail -1; done
done
cp /tmp/shipped.so egonerf_amd/libegonerf_hip.so
