#!/bin/bash
# FEAT_CHAIN: GPU tests of servers that feed servers, chain measurement
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/ab6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flow.py tests/test_gpu_parity.py -x -q -k "feed_servers or chain or launch_order" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python scripts/gpu_r3_chain.py 2048 120 2>&1 | tail -1 | tee $O/chain_2048_120.json
timeout 300 python scripts/gpu_r3_chain.py 10000 600 2>&1 | tail -1 | tee $O/chain_10000_600.json
