#!/bin/bash
# experiment: -mllvm -amdgpu-skip-threshold (when does the compiler put an s_cbranch_execz around a divergent region?)
cd "${GRAFT_REPO_ROOT:-.}"
for thr in default 6 24 48 1000; do
  if [ $thr = default ]; then unset ASYNCFLOW_JIT_EXTRA_FLAGS; else export ASYNCFLOW_JIT_EXTRA_FLAGS="-mllvm -amdgpu-skip-threshold=$thr"; fi
  echo "skip-threshold $thr"
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-diagnostics "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('ms_per_step','kernel_ms','pregen_ms','summary_ms')}, d['parity_spot_check']['ok'])"
done
