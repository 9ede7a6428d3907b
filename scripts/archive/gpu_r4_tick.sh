#!/bin/bash
# Round 4: A/B of the one-region tick_index (AF_TICK_ONE_REGION) on BASELINE config 2, same box, alternating runs.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_tick; mkdir -p $OUT
for rep in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics > $OUT/base_$rep.log 2>&1
  ASYNCFLOW_JIT_EXTRA_FLAGS="-DAF_TICK_ONE_REGION" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics > $OUT/tick_$rep.log 2>&1
done
for f in $OUT/*.log; do echo $f; grep '^{' $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['flow_kernel_ms'], d['parity_spot_check']['ok'], d['config']['flow']['jit_fallbacks'])"; done
ASYNCFLOW_JIT_EXTRA_FLAGS="-DAF_TICK_ONE_REGION" timeout 900 python -m pytest tests/test_gpu_flow.py -m gpu -x -q -k "specialised or prebuilt or grid_corners or far_and_near" 2>&1 | tail -3
