#!/bin/bash
# Compile-level A/B of the plan-specialised af_flow_jit on BASELINE config 2 (built on the box: hipcc, outside the timed region):
# the Philox key schedule per call site (-DAF_PHILOX_KEYS_PER_CALL), the exponential law as a compile-time constant
# (AF_FLOW_DIST_CONST_EXP=1), both.  Interleaved, twice.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05e; mkdir -p $OUT
B="python bench.py --config 2 --steps 8 --warmup 2 --no-cpu-baseline --no-diagnostics"
show() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['ms_per_step'],2), 'flow', round(d['flow_kernel_ms'],2), 'pregen', round(d['pregen_ms'],2), 'summary', round(d['summary_ms'],3), d['parity_spot_check']['ok'], 'jitfb', d['config']['flow']['jit_fallbacks'])"; }
for rep in 1 2; do
  $B > $OUT/fab_base_$rep.log 2>&1; show $OUT/fab_base_$rep.log base
  ASYNCFLOW_JIT_EXTRA_FLAGS="-DAF_PHILOX_KEYS_PER_CALL" $B > $OUT/fab_keys_$rep.log 2>&1; show $OUT/fab_keys_$rep.log keys
  AF_FLOW_DIST_CONST_EXP=1 $B > $OUT/fab_dist_$rep.log 2>&1; show $OUT/fab_dist_$rep.log dist
  AF_FLOW_DIST_CONST_EXP=1 ASYNCFLOW_JIT_EXTRA_FLAGS="-DAF_PHILOX_KEYS_PER_CALL" $B > $OUT/fab_both_$rep.log 2>&1; show $OUT/fab_both_$rep.log both
done
