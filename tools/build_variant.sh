#!/bin/bash
# tools/build_variant.sh <dir> <extra hipcc flags...>: a copy of the package built with extra flags under <dir>/ (for tools/ab_dirs.sh and probes)
d=$1; shift
rm -rf $d; mkdir -p $d/profiles/r05 $d/tools
cp -r egonerf_amd include bench.py oracle $d/ ; cp profiles/r05/pmc_traffic.json $d/profiles/r05/; cp tools/sorted_probe.py $d/tools/
(cd $d && rm -f egonerf_amd/*.so egonerf_amd/*.hash && EGO_EXTRA_FLAGS="$*" python -c "
import sys; sys.path.insert(0,'.')
from egonerf_amd.build import build_library
print(build_library(force=True))" 2>&1 | grep -v "not a recognized" | tail -1)
