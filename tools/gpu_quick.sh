#!/bin/bash
# Short GPU-box session: the -m gpu suite (no -x: every failure is listed) + the bench configs named on the command line.
# Usage: tools/gpu_quick.sh <tag> [pytest args ...]   (EGO_QUICK_BENCH="train erp render" selects bench runs)
TAG=${1:-q}; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${TAG}_build.log 2>&1 || { tail -20 gpurun_out/${TAG}_build.log; exit 1; }
timeout 1800 python -m pytest ${@:-tests} -q -m gpu > gpurun_out/${TAG}_gputest.log 2>&1
tail -15 gpurun_out/${TAG}_gputest.log
for cfg in ${EGO_QUICK_BENCH:-}; do
  timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_$cfg.json 2> gpurun_out/${TAG}_bench_$cfg.err
  python - gpurun_out/${TAG}_bench_$cfg.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["metric"][:40], "ms/step", round(d["ms_per_step"], 4), "value", round(d["value"]), d.get("phases_ms"),
          (d.get("roofline") or {}).get("kernels_ms_serialised"))
except Exception as e:
    print("bench failed:", e, open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
