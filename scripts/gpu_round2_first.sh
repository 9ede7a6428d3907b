set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests/test_gpu_flow.py -x -q > gpurun_out/flow_tests.log 2>&1; echo "rc=$?" >> gpurun_out/flow_tests.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_flow.log 2>&1; echo "rc=$?" >> gpurun_out/bench_flow.log
tail -5 gpurun_out/smoke.log; tail -15 gpurun_out/flow_tests.log; tail -3 gpurun_out/bench_flow.log
