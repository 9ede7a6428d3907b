cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_flow.py -q > gpurun_out/flow_tests.log 2>&1; echo "rc=$?" >> gpurun_out/flow_tests.log
tail -40 gpurun_out/flow_tests.log
