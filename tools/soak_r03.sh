#!/bin/bash
# Final-build soak on the GPU box: determinism of every forward op (8000 repetitions + the full-size batches) and the randomised
# parity campaign (160 cases, both fp16 arithmetics).  -> gpurun_out/soak_*.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 1500 python tools/determinism_check.py 8000 --full > gpurun_out/soak_determinism.txt 2>&1; tail -4 gpurun_out/soak_determinism.txt
timeout 1200 python tools/parity_campaign.py 0 160 > gpurun_out/soak_campaign_f16f8.txt 2>&1; tail -3 gpurun_out/soak_campaign_f16f8.txt
EGO_PREC=f16x3 timeout 1200 python tools/parity_campaign.py 0 160 > gpurun_out/soak_campaign_f16x3.txt 2>&1; tail -3 gpurun_out/soak_campaign_f16x3.txt
