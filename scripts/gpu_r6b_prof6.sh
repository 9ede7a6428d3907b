#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/gensrv_r06b; mkdir -p $OUT
rm -f $OUT/flow_sections_c6_fine.txt
AF_FLOW_PROF=$OUT/flow_sections_c6_fine.txt timeout 600 python bench.py --config 6 --steps 1 --warmup 0 --no-cpu-baseline --no-diagnostics --no-parity-check --separate-summary > $OUT/prof_c6_fine.log 2>&1
tail -22 $OUT/flow_sections_c6_fine.txt; tail -3 $OUT/prof_c6_fine.log | cut -c1-300
