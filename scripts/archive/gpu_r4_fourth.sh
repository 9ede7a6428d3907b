#!/bin/bash
# Round 4, fourth GPU call: general servers with the smaller compact lists, at the sweep sizes that decide flow_wanted().
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_fourth; mkdir -p $OUT
for a in "10000 600" "20000 600" "40000 600" "512 600"; do
  set -- $a
  ( timeout 900 python scripts/gpu_gensrv.py $1 $2 ) > $OUT/gensrv_$1_T$2.json 2>> $OUT/gensrv.err; tail -1 $OUT/gensrv_$1_T$2.json | cut -c1-700
done
