cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do for v in base f16src; do cp egonerf_amd/libvariant_$v.so egonerf_amd/libegonerf_hip.so
EGO_ALLOW_STALE_LIB=1 python bench.py --no-cpu-baseline --no-secondary --cpu-rays 256 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'], 4), 'shade', round(d['roofline']['ms'], 4))"
done; done > gpurun_out/r3_f16src.log 2>&1
cp egonerf_amd/libvariant_f16src.so egonerf_amd/libegonerf_hip.so
EGO_ALLOW_STALE_LIB=1 python bench.py --no-secondary --cpu-rays 1024 --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f16src parity', d['parity'])" >> gpurun_out/r3_f16src.log 2>&1
cat gpurun_out/r3_f16src.log
