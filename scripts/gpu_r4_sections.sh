#!/bin/bash
# Where a wave of af_flow_jit spends its time (FEAT_PROF build, BASELINE config 2) + the plain kernel time.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_sections; mkdir -p $OUT; rm -f $OUT/flow_sections_c2.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > $OUT/base.log 2>&1
AF_FLOW_PROF=$OUT/flow_sections_c2.txt python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-diagnostics --no-parity-check > $OUT/prof.log 2>&1
grep '^{' $OUT/base.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['flow_kernel_ms'], d['parity_spot_check']['ok'])"
tail -16 $OUT/flow_sections_c2.txt
