cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "rc=$?"
grep -v "^{" gpurun_out/bench_default.log | tail -8
python - <<'PY'
import json
l=[x for x in open("gpurun_out/bench_default.log") if x.startswith("{")]
d=json.loads(l[-1])
print(json.dumps({k: d[k] for k in ("value","ms_per_step","kernel_ms","pregen_ms","summary_ms","roofline","cpu_baseline")}, indent=1)[:3500])
PY
