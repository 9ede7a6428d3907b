#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3o; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_flow.py -q -k "pregeneration or lb2_batch or single_server_and_sweep or grid_corners or spike_size" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { tag=$1; shift; python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-diagnostics "$@" > $O/bench_$tag.log 2>&1
  tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'pregen', round(d['pregen_ms'],2), 'flow', round(d['flow_kernel_ms'],2), 'step', round(d['ms_per_step'],2), 'parity', d['parity_spot_check']['ok'], 'value %.3e' % d['value'])" || tail -3 $O/bench_$tag.log; }
run split
AF_PREGEN_VARIANT=rows run rows
run c3 --config 3
run c5 --config 5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt2 -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check > $O/trace.log 2>&1
head -8 $(find /tmp/pt2 -name "*kernel_stats.csv" | head -1) | cut -c1-140
