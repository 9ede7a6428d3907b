#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_rccl; mkdir -p $OUT
( time NCCL_IB_DISABLE=1 NCCL_SOCKET_IFNAME=lo timeout 400 python -m pytest tests/test_gpu_multi.py -m gpu -q -k world_size_one --durations=3 --durations-min=0.5 ) > $OUT/with_env.log 2>&1
grep -E "passed|failed|s call" $OUT/with_env.log | head -4
( time NCCL_DEBUG=INFO timeout 400 python -m pytest tests/test_gpu_multi.py -m gpu -q -s -k world_size_one --durations=3 --durations-min=0.5 ) > $OUT/plain_debug.log 2>&1
grep -E "passed|failed|s call" $OUT/plain_debug.log | head -4
grep -n "NCCL INFO" $OUT/plain_debug.log | head -40 | cut -c1-220
