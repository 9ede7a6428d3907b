#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_cost scripts/microbench/valu_cost.hip > $O/build.log 2>&1
timeout 120 /tmp/valu_cost > $O/valu_cost.txt 2>&1; grep -i "cndmask\|v_add_u32 " $O/valu_cost.txt
( timeout 600 python -m pytest tests/test_gpu_flow.py tests/test_gpu_parity.py -x -q ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for c in 2 5; do python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c$c.log 2>&1
  tail -1 $O/bench_c$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c', 'flow', round(d['flow_kernel_ms'],2), 'step', round(d['ms_per_step'],2), 'pregen', round(d['pregen_ms'],2), 'summary', round(d['summary_ms'],2), 'jit', d['config']['flow']['plan_specialised_kernel'], 'parity', d['parity_spot_check']['ok'], 'value %.3e' % d['value'])"; done
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --generic-kernels > $O/bench_generic.log 2>&1
tail -1 $O/bench_generic.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('generic', 'flow', round(d['flow_kernel_ms'],2), 'step', round(d['ms_per_step'],2))"
rm -f $O/prof_c2.txt; AF_FLOW_PROF=$O/prof_c2.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-check > $O/bench_prof.log 2>&1; tail -n 13 $O/prof_c2.txt
