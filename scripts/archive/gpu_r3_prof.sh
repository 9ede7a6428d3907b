#!/bin/bash
# where a wave of the plan-specialised flow kernel spends its time (FEAT_PROF build, s_memtime per section)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O; rm -f $O/prof_*.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_plain.log 2>&1
AF_FLOW_PROF=$O/prof_c2.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-check > $O/bench_prof.log 2>&1

AF_FLOW_PROF=$O/prof_c5.txt python bench.py --config 5 --scenarios 6250 --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check > $O/bench_prof_c5.log 2>&1
AF_FLOW_PROF=$O/prof_c4.txt python bench.py --config 4 --scenarios 12500 --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check > $O/bench_prof_c4.log 2>&1
for f in plain prof; do tail -1 $O/bench_$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['flow_kernel_ms'], d['ms_per_step'], d['config']['flow']['plan_specialised_kernel'])"; done
tail -n 14 $O/prof_c2.txt; tail -n 14 $O/prof_c2_4096.txt; tail -n 14 $O/prof_c5.txt; tail -n 14 $O/prof_c4.txt
