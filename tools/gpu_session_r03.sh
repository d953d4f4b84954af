cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3_build.log 2>&1
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r3_gputest6.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r3_gputest6.log | tail -12
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err
python -c "
import json; d=json.load(open('gpurun_out/r3_bench2.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['ms'], d['roofline']['other_kernels_ms'], d['roofline']['frac'])
s=d['secondary']
print('train', s['train'].get('ms_per_step'), s['train'].get('cpu_baseline',{}).get('value'), s['train'].get('error'))
print('erp', s['erp'].get('s_per_image'), s['erp'].get('cpu_baseline',{}).get('value'), s['erp'].get('parity'), s['erp'].get('error'))
print('erp_opaque', s['erp_opaque_field'].get('s_per_image'), s['erp_opaque_field'].get('error'))
print('ET', d['roofline']['alt_early_termination'])
"
