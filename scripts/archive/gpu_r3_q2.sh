#!/bin/bash
# EXPERIMENT: where the chain kernel's time goes (results of exp != 0 are wrong on purpose)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3q2; mkdir -p $O
for x in 0 1 2 3; do
AF_PREGEN_EXP=$x timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check > $O/bench_$x.log 2>&1
python - $x $O/bench_$x.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("exp", sys.argv[1], "pregen %.2f flow %.2f ms/step %.2f" % (d["pregen_ms"], d["flow_kernel_ms"], d["ms_per_step"]))
else: print("exp", sys.argv[1], "FAILED", open(sys.argv[2]).read()[-800:])
PY
done
