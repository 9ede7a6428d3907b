#!/bin/bash
# Round 5, second GPU call: the round-5 GPU tests, the general-server workload (bench --config 6: round-at-once station) with
# its section profile, config 5 with tick rows flushed 1 / 8 at a time, the generic server tiers (no scratch object), config 2.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05b; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_flow.py tests/test_gpu_full_batches.py tests/test_gpu_parity.py -m gpu -x -q -k "round or general or several or tiers or thirteen or four_and_five or benchmark_batch or fixtures or full or identical or feed" ) > $OUT/gputests_round5.log 2>&1; echo "rc=$?" >> $OUT/gputests_round5.log; tail -14 $OUT/gputests_round5.log
for c in 6 2; do
  python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_c$c.log 2>&1
  grep '^{' $OUT/bench_c$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config', d['config']['baseline_config'], d['ms_per_step'], d['flow_kernel_ms'], d['config']['flow']['handed_back'], d['config']['flow']['lds_bytes_per_wave'], d['parity_spot_check']['ok'], d['value'])"
done
python bench.py --config 6 --generic-kernels --steps 2 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_c6_generic.log 2>&1
grep '^{' $OUT/bench_c6_generic.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 6 generic', d['ms_per_step'], d['flow_kernel_ms'], d['config']['flow']['handed_back'], d['parity_spot_check']['ok'])"
bash scripts/profile_round5.sh r05b secgensrv
for rows in 1 8 4; do
  ASYNCFLOW_JIT_EXTRA_FLAGS="-DAF_FLUSH_ROWS=$rows" python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_c5_flush$rows.log 2>&1
  grep '^{' $OUT/bench_c5_flush$rows.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 flush rows $rows', d['ms_per_step'], d['flow_kernel_ms'], d['parity_spot_check']['ok'])"
done
python scripts/gpu_chain.py 10000 600 > $OUT/chain_shared_backend_10000_T600.json 2> $OUT/chain.err; cat $OUT/chain_shared_backend_10000_T600.json | cut -c1-900
python bench.py --config 6 --scenarios 40000 --steps 1 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_c6_40000.log 2>&1
grep '^{' $OUT/bench_c6_40000.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 6 x 40000', d['ms_per_step'], d['flow_kernel_ms'], d['config']['flow']['handed_back'], d['parity_spot_check']['ok'])"
ls $OUT
