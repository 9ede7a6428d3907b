#!/bin/bash
# Round 5, first GPU call: the whole-batch parity tests, the VALU calibration, config 5 under the section profile and the PMC
# passes, two ring sizes for config 5, the driver's command.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05a; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_full_batches.py -m gpu -x -q ) > $OUT/gputests_full_batches.log 2>&1; echo "rc=$?" >> $OUT/gputests_full_batches.log; tail -12 $OUT/gputests_full_batches.log
bash scripts/profile_round5.sh r05a cal sec5 c5 sec2
for rows in 64 16; do
  python bench.py --config 5 --flow-ring-rows $rows --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_c5_ring$rows.log 2>&1
  grep '^{' $OUT/bench_c5_ring$rows.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ring', d['config']['flow']['ring_rows'], d['config']['flow']['lds_bytes_per_wave'], d['ms_per_step'], d['flow_kernel_ms'], d['config']['flow']['handed_back'], d['parity_spot_check']['ok'])"
done
bash scripts/profile_round5.sh r05a driver
