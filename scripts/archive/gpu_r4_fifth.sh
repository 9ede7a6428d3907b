#!/bin/bash
# Round 4, fifth GPU call: the full GPU suite on the tree with general servers on the stage-parallel kernel at every size.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_fifth; mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/gputests.log 2>&1; echo "rc=$?" >> $OUT/gputests.log
tail -8 $OUT/gputests.log
