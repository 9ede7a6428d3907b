cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json,sys
tag,path=sys.argv[1],sys.argv[2]
l=[x for x in open(path) if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("%s: value %.3e ms/step %.1f flow_ms %.1f pregen %.1f summary %.1f list %d ring %d fb %s" % (tag, d["value"], d["ms_per_step"], d["flow_kernel_ms"], d["pregen_ms"], d["summary_ms"], d["config"]["flow"]["list_entries"], d["config"]["flow"]["ring_rows"], d["config"]["flow"]["handed_back"]["total"]))
else: print(tag, "FAILED"); print(open(path).read()[-1500:])
PY
}
for w in 2 3 4 5; do
  ASYNCFLOW_HIP_LIB=$PWD/asyncflow_amd/csrc/libasyncflow_hip_w$w.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --generic-kernels > gpurun_out/bench_w$w.log 2>&1
  show "wpe$w" gpurun_out/bench_w$w.log
done
export ASYNCFLOW_HIP_LIB=$PWD/asyncflow_amd/csrc/libasyncflow_hip_w4.so
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --generic-kernels --flow-list-entries 128 > gpurun_out/b1.log 2>&1; show "wpe4 list128" gpurun_out/b1.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --generic-kernels --flow-ring-rows 64 > gpurun_out/b2.log 2>&1; show "wpe4 ring64" gpurun_out/b2.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --generic-kernels --flow-ring-rows -1 > gpurun_out/b3.log 2>&1; show "wpe4 ringHBM" gpurun_out/b3.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --generic-kernels --no-series > gpurun_out/b4.log 2>&1; show "wpe4 no-series" gpurun_out/b4.log
