#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3l; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_flow.py -q -k "pregeneration or lb2_batch or single_server_and_sweep or grid_corners" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in "" 5; do AF_PREGEN_SCEN_PER_WAVE=$v python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-diagnostics > $O/bench_$v.log 2>&1
  tail -1 $O/bench_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pregen variant [$v]', 'pregen', round(d['pregen_ms'],2), 'flow', round(d['flow_kernel_ms'],2), 'step', round(d['ms_per_step'],2), 'parity', d['parity_spot_check']['ok'], 'value %.3e' % d['value'])"; done
for c in 3 4 5; do python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-diagnostics > $O/bench_c$c.log 2>&1
  tail -1 $O/bench_c$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', 'pregen', round(d['pregen_ms'],2), 'flow', round(d['flow_kernel_ms'],2), 'step', round(d['ms_per_step'],2), 'parity', d['parity_spot_check']['ok'], 'value %.3e' % d['value'])"; done
