#!/bin/bash
# Round 6, last session: the driver's command, the same without a compiler (prebuilt kernels of the shipped cache) and smoke() on the tree as it is handed over.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r6c_handover; mkdir -p $OUT
bash scripts/profile_round6.sh r6c_handover driver
ASYNCFLOW_NO_HIPCC=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_without_hipcc_prebuilt_kernels.log 2>&1
grep '^{' $OUT/bench_without_hipcc_prebuilt_kernels.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no hipcc: ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'jit fallbacks', d.get('jit_fallbacks'), 'kernel', d['roofline']['kernel'], 'stale', d['roofline']['binding']['stale'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
