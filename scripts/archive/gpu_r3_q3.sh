#!/bin/bash
# af_arrival_groups: scenarios per workgroup
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3q3; mkdir -p $O
for x in "$@"; do
AF_PREGEN_GROUP=$x timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check > $O/bench_$x.log 2>&1
python - $x $O/bench_$x.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("group", sys.argv[1], "pregen %.2f flow %.2f ms/step %.2f" % (d["pregen_ms"], d["flow_kernel_ms"], d["ms_per_step"]))
else: print("group", sys.argv[1], "FAILED", open(sys.argv[2]).read()[-800:])
PY
done
