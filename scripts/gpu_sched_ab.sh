#!/bin/bash
# A/B of LLVM's AMDGPU scheduling strategies on the plan-specialised kernels (ASYNCFLOW_JIT_EXTRA_FLAGS is part of the cache key;
# the variants are prebuilt in the build container).  One line per (strategy, config): ms per step, flow kernel ms.
OUT=gpurun_out/sched_ab; mkdir -p $OUT
export ASYNCFLOW_NO_HIPCC=1
for c in 2 5 6; do
  for f in default sched-strategy=max-ilp sched-strategy=max-memory-clause sched-strategy=iterative-ilp early-ifcvt=1 schedule-metric-bias=0 schedule-relaxed-occupancy=1; do
    if [ $f = default ]; then unset ASYNCFLOW_JIT_EXTRA_FLAGS; else export ASYNCFLOW_JIT_EXTRA_FLAGS="-mllvm -amdgpu-$f"; fi
    python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check --no-diagnostics > $OUT/c${c}_$f.log 2>&1
    python - "$OUT/c${c}_$f.log" "$c" "$f" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        j = json.loads(l)
        print(sys.argv[2], sys.argv[3], 'ms_per_step', round(j.get('ms_per_step', -1), 2), 'flow_kernel_ms', round(j.get('flow_kernel_ms', -1), 2), 'specialised', j['config']['flow']['plan_specialised_kernel'], 'jit_fallbacks', j['config']['flow']['jit_fallbacks'], 'parity', (j.get('parity_spot_check') or {}).get('ok'))
PY
  done
done
