#!/bin/bash
# what makes the chain kernel slow per message: the optional features (FEAT_ALL) or the levels?
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/ab8; mkdir -p $O
AF_FLOW_PROF=$PWD/$O/prof_chain_lean.txt timeout 300 python scripts/gpu_r3_chain.py 2048 120 2>&1 | tail -1 | tee $O/chain_lean.json
AF_FLOW_FORCE_ALL=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check --scenarios 2048 --horizon 120 > $O/lb2_all.log 2>&1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check --scenarios 2048 --horizon 120 > $O/lb2_lean.log 2>&1
for t in lb2_all lb2_lean; do python - $t $O/$t.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("%-10s events %.3e flow %.2f ms/step %.2f" % (sys.argv[1], d["events_per_step"], d["flow_kernel_ms"], d["ms_per_step"]))
else: print(sys.argv[1], "FAILED", open(sys.argv[2]).read()[-600:])
PY
done
tail -14 $O/prof_chain_lean.txt
