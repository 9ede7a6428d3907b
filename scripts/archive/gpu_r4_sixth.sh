#!/bin/bash
# Round 4, sixth GPU call: general servers after the per-arrival pre-work moved to the arrivals' own lanes + 3-waves/SIMD register budget.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_sixth; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_flow.py -m gpu -x -q -k "endpoints or round_step" ) > $OUT/gputests_gensrv.log 2>&1; echo "rc=$?" >> $OUT/gputests_gensrv.log
tail -4 $OUT/gputests_gensrv.log
for a in "4096 600" "10000 600"; do
  set -- $a
  ( timeout 900 python scripts/gpu_gensrv.py $1 $2 ) > $OUT/gensrv_$1_T$2.json 2>> $OUT/gensrv.err; tail -1 $OUT/gensrv_$1_T$2.json | cut -c1-700
done
