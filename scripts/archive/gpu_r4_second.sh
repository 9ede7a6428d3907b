#!/bin/bash
# Round 4, second GPU call: full GPU suite on the tree with FEAT_CHAIN + planning-only engine, the chain measurement,
# and the bench with hipcc out of reach (prebuilt kernels only).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_second; mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/gputests.log 2>&1; echo "rc=$?" >> $OUT/gputests.log
tail -6 $OUT/gputests.log
( ASYNCFLOW_NO_HIPCC=1 timeout 600 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline ) > $OUT/bench_no_hipcc.log 2>&1; echo "rc=$?" >> $OUT/bench_no_hipcc.log
tail -2 $OUT/bench_no_hipcc.log | cut -c1-1200
( timeout 600 python scripts/gpu_chain.py 2048 120 ) > $OUT/chain_2048_T120.json 2> $OUT/chain.err; tail -1 $OUT/chain_2048_T120.json | cut -c1-1500
( timeout 600 python scripts/gpu_chain.py 10000 600 ) > $OUT/chain_10000_T600.json 2>> $OUT/chain.err; tail -1 $OUT/chain_10000_T600.json | cut -c1-1500
