#!/bin/bash
# Build libegonerf_hip.so once per value of a -D macro and run a check script on each build inside ONE gpurun session.
#   tools/variant_test.sh EGO_SWAP_VARIANT "0 4 7" tools/which_kernel.py
set -e
cd "$(dirname "$0")/.."
MACRO=$1; VALUES=$2; SCRIPT=$3
rm -f egonerf_amd/libvariant_*.so
for v in $VALUES; do
  EGO_EXTRA_FLAGS="-D$MACRO=$v" EGO_LIB_OUT=$PWD/egonerf_amd/libvariant_$v.so python -c "from egonerf_amd.build import build_library as b; b(force=True)" &
done
wait
for v in $VALUES; do test -f egonerf_amd/libvariant_$v.so || { echo "variant $v failed to build"; exit 1; }; done
/usr/local/graft/bin/gpurun --timeout 900 -- "for v in $VALUES; do cp egonerf_amd/libvariant_\$v.so egonerf_amd/libegonerf_hip.so; echo \"== $MACRO=\$v\"; EGO_ALLOW_STALE_LIB=1 $SCRIPT 2>&1 | tail -4; done" 2>&1 | tail -40
rm -f egonerf_amd/libvariant_*.so
python -c "import __graft_entry__ as g; g.build()" | tail -1
