for i in 1 2 3; do EGO_ALLOW_STALE_LIB=1 EGO_SKIP_SELFTEST=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), 'shade', round(d['roofline']['ms'],4), 'parity', d.get('parity'))"; done
