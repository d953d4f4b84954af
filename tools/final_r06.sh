#!/bin/bash
# Final-tree records of round 6 (on the GPU box): build + smoke, full -m gpu suite, the driver's bench command, profile, kernel tables, soak.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/f6_build_smoke.log 2>&1; tail -1 gpurun_out/f6_build_smoke.log
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/f6_gputest.log 2>&1; tail -3 gpurun_out/f6_gputest.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/f6_bench_line.json 2> gpurun_out/f6_bench_stderr.log
echo "bench rc=$? bytes $(tail -1 gpurun_out/f6_bench_line.json | wc -c)"; cp gpurun_out/bench_full.json gpurun_out/f6_bench_full.json
tail -1 gpurun_out/f6_bench_line.json | head -c 700; echo
bash tools/profile_r06.sh f6 r06 > gpurun_out/f6_profile_stdout.txt 2>&1; tail -2 gpurun_out/f6_profile_stdout.txt
EGO_SKIP_SELFTEST=1 python tools/pcie_inclusive.py > gpurun_out/f6_pcie.json 2>/dev/null; cat gpurun_out/f6_pcie.json
tools/train_kernels.sh > gpurun_out/f6_train_kernels.txt 2>&1; head -12 gpurun_out/f6_train_kernels.txt
(echo "== walk form (default)"; tools/sorted_kernels.sh 2>&1 | grep "k_\|ms"; echo "== r05 form (EGO_SORTED_WALK=0)"; tools/sorted_kernels.sh EGO_SORTED_WALK=0 2>&1 | grep "k_\|ms") > gpurun_out/f6_sorted_kernels.txt; grep -c k_ gpurun_out/f6_sorted_kernels.txt
bash tools/soak_r06.sh > gpurun_out/f6_soak.txt 2>&1; tail -3 gpurun_out/f6_soak.txt
