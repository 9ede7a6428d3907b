#!/bin/bash
# Round 3, re-entry: the GPU suite, the rocprofv3 evidence of the final kernels (af_arrival_groups + af_flow_jit), the default
# bench line with its CPU baseline, and the other configs -- most important first, each under its own timeout.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/final1; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log
tail -4 $O/gputests.log
timeout 900 bash scripts/profile_round3.sh r03 > $O/profile.log 2>&1; echo "profile rc=$?"
head -12 gpurun_out/prof_r03/kernel_stats_trace.csv | cut -c1-160
( time timeout 400 python bench.py ) > $O/bench_default.log 2>&1; echo "bench rc=$?"
show() { python - "$1" "$2" <<'PY'
import json,sys
tag,path=sys.argv[1],sys.argv[2]
l=[x for x in open(path) if x.startswith("{")]
if l:
    d=json.loads(l[-1])
    print("%-10s value %.3e ms/step %.2f flow_ms %.2f pregen %.2f summary %.2f parity %s" % (tag, d["value"], d["ms_per_step"], d["flow_kernel_ms"], d["pregen_ms"], d["summary_ms"], d.get("parity_spot_check")))
else: print(tag, "FAILED"); print(open(path).read()[-1500:])
PY
}
show default $O/bench_default.log
for c in 3 5 4; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-diagnostics > $O/bench_c$c.log 2>&1; show c$c $O/bench_c$c.log
done
