#!/bin/bash
# A/B of plan-specialised builds of af_flow_jit on BASELINE config 2, same box, interleaved:
#   bash scripts/gpu_ab_flags.sh <tag> "<flags of variant 1>" "<flags of variant 2>" ...     ("" = the tree as it is)
# then the flow GPU tests that use specialised builds with the LAST variant's flags.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/ab_$TAG; mkdir -p $OUT
for rep in 1 2; do
  i=0
  for flags in "$@"; do
    ASYNCFLOW_JIT_EXTRA_FLAGS="$flags" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics > $OUT/v${i}_$rep.log 2>&1
    i=$((i+1))
  done
done
i=0
for flags in "$@"; do
  for rep in 1 2; do
    printf "%-40s " "[$flags]"; grep '^{' $OUT/v${i}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['flow_kernel_ms'],2), d['parity_spot_check']['ok'], d['config']['flow']['jit_fallbacks'])"
  done
  i=$((i+1))
done
last="${@: -1}"
ASYNCFLOW_JIT_EXTRA_FLAGS="$last" timeout 900 python -m pytest tests/test_gpu_flow.py -m gpu -x -q -k "specialised or prebuilt or grid_corners or far_and_near or sweep_over" 2>&1 | tail -2
