#!/bin/bash
# Round 6: af_arrival_groups' workgroups launched heaviest group first (sweeps over the load) -- interleaved A/B on configs 4 and 3.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pregen_r06; mkdir -p $OUT
line() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'pregen %.2f' % d['pregen_ms'], 'value %.4g' % d['value'], 'parity', d['parity_spot_check']['ok'])"; }
for c in 4 3; do
  for rep in 1 2; do
    python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > $OUT/bench_c${c}_heaviest_first_$rep.log 2>&1; line $OUT/bench_c${c}_heaviest_first_$rep.log "config $c heaviest group first"
    AF_PREGEN_ORDER_OFF=1 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > $OUT/bench_c${c}_index_order_$rep.log 2>&1; line $OUT/bench_c${c}_index_order_$rep.log "config $c index order         "
  done
done
( time timeout 1500 python -m pytest tests/test_gpu_full_batches.py tests/test_gpu_parity.py -m gpu -q -k "config_4 or arrival or grid or sweep" ) > $OUT/gputests_grids.log 2>&1; tail -6 $OUT/gputests_grids.log
