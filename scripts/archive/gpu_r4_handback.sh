#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_handback; mkdir -p $OUT
python scripts/gpu_handback.py > $OUT/handback.json 2> $OUT/err.log; tail -1 $OUT/handback.json; tail -3 $OUT/err.log
