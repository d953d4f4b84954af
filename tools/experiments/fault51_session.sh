#!/bin/bash
# runs ON the GPU box (gpurun) after tools/experiments/fault51_asm_patch.py has put libvariant_{f0,p1,p2,p3,p4}.so
# (and, optionally, libvariant_f1.so = the faulty form built with -mllvm -amdgpu-waitcnt-forcezero) into egonerf_amd/.
# Alternates the builds twice under tools/flaky_probe.py and restores the shipped library.
cd $GRAFT_REPO_ROOT
cp egonerf_amd/libegonerf_hip.so /tmp/shipped.so
for rep in 1 2; do
  for v in f0 p1 f1 p2 p3 p4; do
    [ -f egonerf_amd/libvariant_$v.so ] || continue
    cp egonerf_amd/libvariant_$v.so egonerf_amd/libegonerf_hip.so
    echo "== $v (rep $rep)"; timeout 300 python tools/flaky_probe.py 3000 2>&1 | tail -2 | cut -c1-160
  done
done
cp /tmp/shipped.so egonerf_amd/libegonerf_hip.so
echo "== shipped"; timeout 300 python tools/flaky_probe.py 3000 2>&1 | tail -2 | cut -c1-160
