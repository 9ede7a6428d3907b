#!/bin/bash
# Round 5, third GPU call: the general-server workload after the LDS-rank form of the round-at-once solver, config 5 register budgets.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05c; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_flow.py -m gpu -x -q -k "round_at_a_time or benchmark_batch or several_endpoints or tiers_of_general or round_step" ) > $OUT/gputests_gensrv.log 2>&1; echo "rc=$?" >> $OUT/gputests_gensrv.log; tail -6 $OUT/gputests_gensrv.log
python bench.py --config 6 --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_c6.log 2>&1
grep '^{' $OUT/bench_c6.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 6', d['ms_per_step'], d['flow_kernel_ms'], d['config']['flow']['handed_back'], d['config']['flow']['lds_bytes_per_wave'], d['parity_spot_check']['ok'], d['value'])"
bash scripts/profile_round5.sh r05c secgensrv
for wpe in 2 3; do
  ASYNCFLOW_JIT_EXTRA_FLAGS="-DAF_FLOW_WPE=$wpe" python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_c5_wpe$wpe.log 2>&1
  grep '^{' $OUT/bench_c5_wpe$wpe.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 waves_per_eu $wpe', d['ms_per_step'], d['flow_kernel_ms'], d['parity_spot_check']['ok'])"
done
ASYNCFLOW_JIT_EXTRA_FLAGS="-DAF_FLOW_WPE=3" python bench.py --config 6 --steps 2 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_c6_wpe.log 2>&1
grep '^{' $OUT/bench_c6_wpe.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 6 again', d['ms_per_step'], d['flow_kernel_ms'], d['parity_spot_check']['ok'])"
ls $OUT
