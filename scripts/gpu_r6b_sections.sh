#!/bin/bash
# Round 6 (second session): where a wave of af_flow_jit spends its time on configs 5, 6 and 3 (FEAT_PROF builds), and how
# config 5 answers to the tick ring's length (--flow-ring-rows) and the lists' length.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/sections_r06b; mkdir -p $OUT
line() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'value %.4g' % d['value'])"; }
for c in 5 6 3; do
  rm -f $OUT/flow_sections_c$c.txt
  AF_FLOW_PROF=$OUT/flow_sections_c$c.txt timeout 600 python bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-diagnostics --no-parity-check --separate-summary > $OUT/prof_c$c.log 2>&1
  echo "=== config $c"; tail -16 $OUT/flow_sections_c$c.txt
done
timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check --separate-summary > $OUT/c5_base.log 2>&1; line $OUT/c5_base.log "c5 base       "
for rr in 16 64 -1; do
  timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check --separate-summary --flow-ring-rows $rr > $OUT/c5_ring$rr.log 2>&1; line $OUT/c5_ring$rr.log "c5 ring rows $rr"
done
for le in 64 256; do
  timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check --separate-summary --flow-list-entries $le > $OUT/c5_list$le.log 2>&1; line $OUT/c5_list$le.log "c5 list entries $le"
done
