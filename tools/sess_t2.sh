cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
for v in base ldsang; do cp egonerf_amd/libvariant_$v.so egonerf_amd/libegonerf_hip.so; echo -n "$v: "; python tools/march_timing.py 2>&1 | tail -1; done
done
cp egonerf_amd/libvariant_base.so egonerf_amd/libegonerf_hip.so
timeout 1500 python -m pytest tests/test_hip_train.py tests/test_hip_train_extras.py tests/test_hip_wgrad.py tests/test_hip_convergence.py -q -m gpu > gpurun_out/t2_gputest.log 2>&1; tail -8 gpurun_out/t2_gputest.log
for i in 1 2; do python bench.py --config train --no-cpu-baseline --no-secondary > gpurun_out/t2_bench_train_$i.json 2> gpurun_out/t2_bench_train.err; python -c "
import json,sys
d=json.loads(open('gpurun_out/t2_bench_train_$i.json').read().strip().splitlines()[-1]); print('train ms/step', d['ms_per_step'], d['phases_ms']); print(d['roofline'].get('kernels_ms_serialised'))"; done
