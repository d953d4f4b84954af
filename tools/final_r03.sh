#!/bin/bash
# Round-end check on the GPU box: build + smoke, the whole -m gpu suite, the driver's bench command (-> gpurun_out/final_bench_line.json)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/final_build_smoke.log 2>&1; tail -2 gpurun_out/final_build_smoke.log
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/final_gputest.log 2>&1; tail -3 gpurun_out/final_gputest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench_line.json 2> gpurun_out/final_bench_stderr.log
python - <<'PY'
import json
l = json.loads(open('gpurun_out/final_bench_line.json').read().strip().splitlines()[-1])
r = l['roofline']
print('render', l['value'], l['ms_per_step'], 'frac', r['frac'], 'shade ms', r['ms'], 'stale', r['inputs']['stale_vs_current_sources'], 'hbm_counter_frac', r['hbm_counter_frac'])
print('alt', {k: (round(v['ms_per_step'], 4), v['max_abs_rgb_err']) for k, v in r['alt_precision'].items()})
for k, v in l['secondary'].items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('step_mode'), v.get('eager_ms_per_step'), (v.get('roofline') or {}).get('frac'), (v.get('cpu_baseline') or {}).get('value'))
print('cpu', l['cpu_baseline']['value'], 'parity', l['parity']['max_abs_rgb_err'])
PY
