#!/bin/bash
# Round 4, first GPU call: the GPU suite and the driver's exact bench command on the restored tree.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_first; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/gputests.log 2>&1; echo "rc=$?" >> $OUT/gputests.log
tail -6 $OUT/gputests.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd.log 2>&1; echo "rc=$?" >> $OUT/bench_driver_cmd.log
tail -3 $OUT/bench_driver_cmd.log | cut -c1-1500
