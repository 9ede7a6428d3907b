#!/bin/bash
# split arrival pre-generation (af_pregen.hpp): parity + A/B against the row kernel + per-kernel times
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flow.py -x -q -k "pregeneration" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
show() { python - "$1" "$2" <<'PY'
import json,sys
tag,path=sys.argv[1],sys.argv[2]
l=[x for x in open(path) if x.startswith("{")]
if l:
    d=json.loads(l[-1])
    print("%-10s value %.3e ms/step %.2f flow_ms %.2f pregen %.2f summary %.2f parity %s" % (tag, d["value"], d["ms_per_step"], d["flow_kernel_ms"], d["pregen_ms"], d["summary_ms"], d.get("parity_spot_check")))
else: print(tag, "FAILED"); print(open(path).read()[-2500:])
PY
}
AF_PREGEN_MODE=rows timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > $O/bench_rows.log 2>&1; show rows $O/bench_rows.log
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > $O/bench_split.log 2>&1; show split $O/bench_split.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check > $O/bench_under_trace.log 2>&1
cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_trace.csv; head -8 $O/kernel_stats_trace.csv | cut -c1-200
