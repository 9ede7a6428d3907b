#!/bin/bash
# Round 6: the one-call step against run + summarize on the grid sweeps (configs 3 / 4: launched heaviest first), then the whole GPU suite.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/fused_r06; mkdir -p $OUT
line() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['summary']; print('$2', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'pregen %.2f' % d['pregen_ms'], 'summary after %.2f beside %.2f over %d scenarios' % (s['ms'], s['beside_ms'], s['overlapped_scenarios']), 'value %.4g' % d['value'], 'parity', d['parity_spot_check']['ok'])"; }
for c in 3 4; do
  for rep in 1 2; do
    python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > $OUT/bench_c${c}_one_call_$rep.log 2>&1; line $OUT/bench_c${c}_one_call_$rep.log "config $c one call    "
    python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics --separate-summary > $OUT/bench_c${c}_two_calls_$rep.log 2>&1; line $OUT/bench_c${c}_two_calls_$rep.log "config $c two calls   "
  done
done
( time timeout 3000 python -m pytest tests -m gpu -q --durations=12 ) > $OUT/gputests_full.log 2>&1; tail -25 $OUT/gputests_full.log
