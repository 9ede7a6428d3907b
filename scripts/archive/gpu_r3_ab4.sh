#!/bin/bash
# heaviest-first launch order of the stage-parallel kernel on sweeps over the load: tests, configs 3 / 4 with and without
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/ab4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flow.py -x -q -k "pregeneration or grid" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics "$@" > $O/$tag.log 2>&1
  python - $tag $O/$tag.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("%-10s value %.3e pregen %.2f flow %.2f ms/step %.2f parity %s" % (sys.argv[1], d["value"], d["pregen_ms"], d["flow_kernel_ms"], d["ms_per_step"], d.get("parity_spot_check",{}).get("ok")))
else: print(sys.argv[1], "FAILED", open(sys.argv[2]).read()[-600:])
PY
}
run c3_off AF_FLOW_ORDER_OFF=1 -- --config 3
run c3_on X=1 -- --config 3
run c4_on X=1 -- --config 4
