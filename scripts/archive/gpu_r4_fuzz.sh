#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_fuzz; mkdir -p $OUT
( time timeout 1500 python scripts/gpu_fuzz_f3.py 40 ) > $OUT/gpu_fuzz_f3.json 2> $OUT/gpu_fuzz_f3.err; echo "rc=$?"; tail -1 $OUT/gpu_fuzz_f3.json | cut -c1-1500; tail -5 $OUT/gpu_fuzz_f3.err
