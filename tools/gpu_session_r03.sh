cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3_build.log 2>&1
timeout 1200 python -m pytest tests/test_model_shapes.py tests/test_abi_and_host.py -q > gpurun_out/r3_gputest7.log 2>&1
tail -40 gpurun_out/r3_gputest7.log
