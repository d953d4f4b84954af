#!/bin/bash
# Final-build soak on the GPU box: determinism of every forward op (3000 repetitions + the full-size batches, all split arithmetics) and
# the randomised parity campaign in the default arithmetic (f16f6), seeds 0-3 x 160 cases.  -> gpurun_out/soak4_*.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 1500 python tools/determinism_check.py 3000 --full > gpurun_out/soak4_determinism.txt 2>&1; tail -6 gpurun_out/soak4_determinism.txt
for seed in 0 1 2 3; do
  EGO_PREC=f16f6 timeout 1200 python tools/parity_campaign.py $seed 160 > gpurun_out/soak4_campaign_f16f6_seed$seed.txt 2>&1; tail -2 gpurun_out/soak4_campaign_f16f6_seed$seed.txt
done
