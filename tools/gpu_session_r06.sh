#!/bin/bash
# One GPU-box session (via gpurun), round 6: build check + smoke, then the commands given as arguments, each with its own timeout, output
# tails on stdout and full logs under gpurun_out/<tag>_<n>.log
TAG=${1:-s}; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/${TAG}_build_smoke.log 2>&1 || { tail -20 gpurun_out/${TAG}_build_smoke.log; exit 1; }
tail -1 gpurun_out/${TAG}_build_smoke.log
n=0
for cmd in "$@"; do
  n=$((n + 1))
  echo "== [$n] $cmd"
  timeout 1700 bash -c "$cmd" > gpurun_out/${TAG}_$n.log 2>&1
  echo "rc=$?"; tail -${TAIL:-25} gpurun_out/${TAG}_$n.log
done
