#!/bin/bash
# One GPU-box session (via gpurun): build check + smoke, optional -m gpu suite, the driver's bench command, extra commands.
# Usage: tools/gpu_session_r05.sh <tag> [tests|notests|nobench|only] [extra command ...]  -> gpurun_out/<tag>_*
#   tests = full -m gpu suite + bench; notests = bench only; nobench = suite only; only = just the extra commands
TAG=${1:-s}; DO=${2:-tests}; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/${TAG}_build_smoke.log 2>&1 || { tail -20 gpurun_out/${TAG}_build_smoke.log; exit 1; }
tail -1 gpurun_out/${TAG}_build_smoke.log
if [ "$DO" = "tests" ] || [ "$DO" = "nobench" ]; then
  timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gputest.log 2>&1
  tail -8 gpurun_out/${TAG}_gputest.log
fi
if [ "$DO" = "tests" ] || [ "$DO" = "notests" ]; then
  timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log
  echo "bench rc=$? line bytes: $(tail -1 gpurun_out/${TAG}_bench_line.json | wc -c)"
  tail -1 gpurun_out/${TAG}_bench_line.json | head -c 1500; echo
  cp gpurun_out/bench_full.json gpurun_out/${TAG}_bench_full.json 2>/dev/null
fi
for cmd in "$@"; do
  echo "== $cmd"
  timeout 1500 bash -c "$cmd" 2>&1 | tail -30
done
