#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_cost scripts/microbench/valu_cost.hip > $O/build.log 2>&1
timeout 120 /tmp/valu_cost > $O/valu_cost.txt 2>&1; cat $O/valu_cost.txt
for c in 2 3 4 5; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c$c.log 2>&1
  tail -1 $O/bench_c$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', 'flow', round(d['flow_kernel_ms'],2), 'step', round(d['ms_per_step'],2), 'pregen', round(d['pregen_ms'],2), 'summary', round(d['summary_ms'],2), 'jit', d['config']['flow']['plan_specialised_kernel'], 'parity', d['parity_spot_check']['ok'], 'value %.3e' % d['value'])"; done
