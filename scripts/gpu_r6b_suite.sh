#!/bin/bash
# Round 6 (second session): bench lines of every config on the tree + the whole GPU suite.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/suite_r06b; mkdir -p $OUT
line() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'value %.4g' % d['value'], 'parity', d.get('parity_spot_check',{}).get('ok'), 'jit_fallbacks', d['config']['flow']['jit_fallbacks'])"; }
for c in 2 3 4 5 6; do
  timeout 900 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > $OUT/bench_c$c.log 2>&1; line $OUT/bench_c$c.log "config $c one call "
  timeout 900 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics --separate-summary > $OUT/bench_c${c}_two_calls.log 2>&1; line $OUT/bench_c${c}_two_calls.log "config $c two calls"
done
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $OUT/gputests.log 2>&1; tail -6 $OUT/gputests.log
