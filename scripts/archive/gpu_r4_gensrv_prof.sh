#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_gsprof; mkdir -p $OUT; rm -f $OUT/sections.txt
python scripts/gpu_gensrv_prof.py 2304 120 > $OUT/plain.log 2>&1; tail -1 $OUT/plain.log
AF_FLOW_PROF=$OUT/sections.txt python scripts/gpu_gensrv_prof.py 2304 120 > $OUT/prof.log 2>&1; tail -1 $OUT/prof.log
tail -16 $OUT/sections.txt
