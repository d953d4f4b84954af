#!/bin/bash
# One GPU-box session (via gpurun): build check + smoke, optional -m gpu suite, the driver's bench command, extra commands.
# Usage: tools/gpu_session_r04.sh <tag> [tests|notests] [extra command ...]  -> gpurun_out/<tag>_*
TAG=${1:-s}; DO_TESTS=${2:-tests}; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/${TAG}_build_smoke.log 2>&1 || { tail -20 gpurun_out/${TAG}_build_smoke.log; exit 1; }
tail -1 gpurun_out/${TAG}_build_smoke.log
if [ "$DO_TESTS" = "tests" ]; then
  timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gputest.log 2>&1
  tail -8 gpurun_out/${TAG}_gputest.log
fi
if [ "$DO_TESTS" != "nobench" ]; then
  timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log
  python tools/line_brief.py gpurun_out/${TAG}_bench_line.json 2>/dev/null || head -c 400 gpurun_out/${TAG}_bench_line.json; echo
fi
for cmd in "$@"; do
  echo "== $cmd"
  timeout 1200 bash -c "$cmd" 2>&1 | tail -25
done
