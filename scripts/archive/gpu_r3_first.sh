#!/bin/bash
# round 3, first GPU visit: tests, then the bench with the plan-specialised and with the generic flow kernel
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r3a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
AF_DEBUG=1 python bench.py --steps 10 --warmup 2 > gpurun_out/r3a/bench_jit.log 2>&1
python bench.py --steps 10 --warmup 2 --generic-kernels --no-cpu-baseline > gpurun_out/r3a/bench_generic.log 2>&1
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --scenarios 8192 > gpurun_out/r3a/bench_jit_8192.log 2>&1
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-series > gpurun_out/r3a/bench_jit_noseries.log 2>&1
for c in 3 4 5; do python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench_c$c.log 2>&1; done
tail -3 gpurun_out/r3a/pytest.log
for f in gpurun_out/r3a/bench_*.log; do echo "== $f"; tail -1 $f | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'pregen_ms', 'flow_kernel_ms', 'summary_ms')}, d['config']['flow'], d.get('parity_spot_check'))
except Exception as e:
    print('unparsable', e)
"; done
