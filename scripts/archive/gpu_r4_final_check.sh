#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_final; mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/gputests.log 2>&1; echo "rc=$?" >> $OUT/gputests.log
tail -4 $OUT/gputests.log
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd.log 2>&1; echo "rc=$?" >> $OUT/bench_driver_cmd.log
grep '^{' $OUT/bench_driver_cmd.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['kernel_ms'], d['roofline']['binding_detail'].get('stale'), d['config']['flow']['jit_fallbacks'], d['parity_spot_check']['ok'])"
( AF_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/bench_forced_dist.log 2>&1; echo "rc=$?" >> $OUT/bench_forced_dist.log
grep '^{' $OUT/bench_forced_dist.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced dist:', d['ms_per_step'], d['gather_path'], d['gather_fallback'])"
