#!/bin/bash
# quick bench lines (two calls: the kernel alone) of the configs given as arguments
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/quick_r06b; mkdir -p $OUT
line() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'value %.4g' % d['value'], 'parity', d.get('parity_spot_check',{}).get('ok'), 'jit_fallbacks', d['config']['flow']['jit_fallbacks'])"; }
for c in "$@"; do
  for rep in 1 2; do
    timeout 900 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics --separate-summary > $OUT/bench_c${c}_$rep.log 2>&1; line $OUT/bench_c${c}_$rep.log "config $c two calls"
  done
done
