#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_lcchain; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_flow.py -m gpu -x -q -k "tiers or feed_servers or least" ) > $OUT/gputests.log 2>&1; echo "rc=$?" >> $OUT/gputests.log
tail -4 $OUT/gputests.log
