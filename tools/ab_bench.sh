#!/bin/bash
# A/B of two builds of libegonerf_hip.so inside ONE gpurun session (boxes differ by +-3 %, so numbers from different calls
# cannot resolve a 1 % kernel change).  Usage, from the repo root with the candidate change in the working tree:
#   tools/ab_bench.sh            -> builds HEAD ("old", via git stash) and the working tree ("new"), then alternates bench.py
set -e
cd "$(dirname "$0")/.."
build() { python -c "import __graft_entry__ as g; g.build()" | tail -1; }
build; cp egonerf_amd/libegonerf_hip.so /tmp/ab_new.so
git stash -q; build; cp egonerf_amd/libegonerf_hip.so egonerf_amd/libvariant_old.so; git stash pop -q
cp /tmp/ab_new.so egonerf_amd/libvariant_new.so; build
/usr/local/graft/bin/gpurun --timeout 1200 -- 'for rep in 1 2 3; do for v in old new; do cp egonerf_amd/libvariant_$v.so egonerf_amd/libegonerf_hip.so; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"'"'"'$v'"'"'\", round(d[\"ms_per_step\"], 4), \"shade\", round(d[\"roofline\"][\"ms\"], 4), d[\"roofline\"][\"other_kernels_ms\"])"; done; done' 2>&1 | tail -7
rm -f egonerf_amd/libvariant_old.so egonerf_amd/libvariant_new.so
