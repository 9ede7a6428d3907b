#!/bin/bash
# neighbours_alike policy: the new GPU test, the pre-generation tests, configs 3 / 4 with the default policy
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/ab3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flow.py -x -q -k "pregeneration or grid" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for c in 3 4; do
  timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics > $O/c$c.log 2>&1
  python - c$c $O/c$c.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("%-6s value %.3e pregen %.2f flow %.2f ms/step %.2f parity %s" % (sys.argv[1], d["value"], d["pregen_ms"], d["flow_kernel_ms"], d["ms_per_step"], d.get("parity_spot_check",{}).get("ok")))
else: print(sys.argv[1], "FAILED", open(sys.argv[2]).read()[-600:])
PY
done
