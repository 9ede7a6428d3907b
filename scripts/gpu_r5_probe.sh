#!/bin/bash
# What would a generator station that draws its own gaps cost?    A/B on BASELINE config 2, same box,
# interleaved: the tree as it is / the same kernel doing the fused generator's work on top (results discarded: AF_FUSED_ARRIVAL_PROBE).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05d; mkdir -p $OUT
for rep in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics > $OUT/fused_probe_off_$rep.log 2>&1
  ASYNCFLOW_JIT_EXTRA_FLAGS="-DAF_FUSED_ARRIVAL_PROBE" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics > $OUT/fused_probe_on_$rep.log 2>&1
done
for f in off_1 on_1 off_2 on_2; do printf "%-8s" $f; grep '^{' $OUT/fused_probe_$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'flow', round(d['flow_kernel_ms'],2), 'pregen', round(d['pregen_ms'],2), d['parity_spot_check']['ok'], d['config']['flow']['jit_fallbacks'])"; done
