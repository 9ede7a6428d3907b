#!/bin/bash
# Interleaved A/B of plan-specialised builds (ASYNCFLOW_JIT_EXTRA_FLAGS variants, prebuilt in the build container) over several configs:
#   bash scripts/gpu_r6b_ab.sh <tag> "<configs>" "<flags of variant 0>" "<flags of variant 1>" ...
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=$1; CFGS=$2; shift; shift
OUT=gpurun_out/ab_$TAG; mkdir -p $OUT
for c in $CFGS; do
  for rep in 1 2; do
    i=0
    for flags in "$@"; do
      ASYNCFLOW_JIT_EXTRA_FLAGS="$flags" timeout 600 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-diagnostics --separate-summary > $OUT/c${c}_v${i}_$rep.log 2>&1
      i=$((i+1))
    done
  done
  i=0
  for flags in "$@"; do
    for rep in 1 2; do
      printf "config %s %-44s " $c "[$flags]"; grep '^{' $OUT/c${c}_v${i}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['flow_kernel_ms'],2), d['parity_spot_check']['ok'], d['config']['flow']['jit_fallbacks'])"
    done
    i=$((i+1))
  done
done
