#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_gsjit; mkdir -p $OUT
for a in "512 600" "10000 600" "40000 600"; do
  set -- $a
  ( timeout 900 python scripts/gpu_gensrv.py $1 $2 ) > $OUT/gensrv_jit_$1_T$2.json 2>> $OUT/err.log; tail -1 $OUT/gensrv_jit_$1_T$2.json | cut -c1-1100
done
