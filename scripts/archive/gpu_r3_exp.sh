#!/bin/bash
# A/B of JIT build flags for the flow kernel: bash scripts/gpu_r3_exp.sh <outdir> "<flags A>" "<flags B>" ...
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
i=0
for flags in "$@"; do
  export ASYNCFLOW_JIT_EXTRA_FLAGS="$flags"
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_$i.log 2>&1
  tail -1 $O/bench_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$flags]', 'flow', round(d['flow_kernel_ms'],2), 'step', round(d['ms_per_step'],2), 'jit', d['config']['flow']['plan_specialised_kernel'], 'parity', d['parity_spot_check']['ok'])" || tail -5 $O/bench_$i.log
  i=$((i+1))
done
