#!/bin/bash
# Round 6: af_engine_run_summarized (analyzer beside the stage-parallel kernel's last residency round) -- the GPU test, then
# interleaved A/B of the bench step with the one call against run + summarize, configs 2 / 5 / 6.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/fused_r06; mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_gpu_analyzer.py -m gpu -q --durations=5 ) > $OUT/gputests_analyzer.log 2>&1; tail -12 $OUT/gputests_analyzer.log
line() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['summary']; print('$2', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'pregen %.2f' % d['pregen_ms'], 'summary after %.2f beside %.2f over %d scenarios' % (s['ms'], s['beside_ms'], s['overlapped_scenarios']), 'value %.4g' % d['value'], 'parity', d['parity_spot_check']['ok'])"; }
for c in 2 5 6; do
  for rep in 1 2; do
    python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics > $OUT/bench_c${c}_one_call_$rep.log 2>&1; line $OUT/bench_c${c}_one_call_$rep.log "config $c one call    "
    python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics --separate-summary > $OUT/bench_c${c}_two_calls_$rep.log 2>&1; line $OUT/bench_c${c}_two_calls_$rep.log "config $c two calls   "
  done
done
AF_DEBUG=1 python bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline --no-diagnostics --no-parity-check 2>&1 | grep "two parts" | head -3
