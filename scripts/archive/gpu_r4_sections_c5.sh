#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_sections_c5; mkdir -p $OUT; rm -f $OUT/*.txt
for c in 5 4 3; do
AF_FLOW_PROF=$OUT/flow_sections_c$c.txt python bench.py --config $c --scenarios 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-diagnostics --no-parity-check > $OUT/prof_c$c.log 2>&1
tail -14 $OUT/flow_sections_c$c.txt
done
