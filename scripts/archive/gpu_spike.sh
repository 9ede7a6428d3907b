cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_flow.py -x -q -m gpu ) > gpurun_out/flow_tests.log 2>&1; tail -5 gpurun_out/flow_tests.log
AF_DEBUG=1 timeout 600 python scripts/spike_examples.py 2048 2>&1 | grep -v "^\[af\] flow second chance.*flags" | tail -30
bash scripts/gpu_ab.sh "|" "|--config 4"
