#!/bin/bash
# One GPU-box session (via gpurun): build check, the -m gpu suite, the driver's bench command, then the round's profile passes.
# Usage: tools/gpu_session_r03.sh <tag>   -> gpurun_out/{<tag>_gputest.log, <tag>_bench_line.json, prof_<tag>/...}
TAG=${1:-v2}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/${TAG}_build_smoke.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gputest.log 2>&1
tail -5 gpurun_out/${TAG}_gputest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log
head -c 600 gpurun_out/${TAG}_bench_line.json; echo
timeout 1500 bash tools/profile_r03.sh $TAG r03 > gpurun_out/${TAG}_profile.log 2>&1
tail -3 gpurun_out/${TAG}_profile.log
