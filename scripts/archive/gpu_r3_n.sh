#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3n; mkdir -p $O
run() { tag=$1; shift; python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-diagnostics "$@" > $O/bench_$tag.log 2>&1
  tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'pregen', round(d['pregen_ms'],2), 'flow', round(d['flow_kernel_ms'],2), 'step', round(d['ms_per_step'],2), 'lds', d['config']['flow']['lds_bytes_per_wave'], 'rows', d['config']['flow']['ring_rows'], 'parity', d['parity_spot_check']['ok'], 'value %.3e' % d['value'])" || tail -3 $O/bench_$tag.log; }
run default
AF_FLOW_NO_DIST_CONST=1 run nodist
AF_FLOW_NO_DIST_CONST=1 run nodist_noseries --no-series
run c5 --config 5
AF_FLOW_NO_DIST_CONST=1 run c5_nodist --config 5
