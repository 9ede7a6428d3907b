# A/B of library builds / bench flags on the GPU box: bash scripts/gpu_ab.sh "<lib suffix>|<bench flags>" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json,sys
tag,path=sys.argv[1],sys.argv[2]
l=[x for x in open(path) if x.startswith("{")]
if l:
    d=json.loads(l[-1]); f=d["config"]["flow"]
    print("%-40s value %.3e ms/step %.1f flow_ms %.1f pregen %.1f summary %.1f list %d ring %d lds %d fb %s frac %.3f" % (tag, d["value"], d["ms_per_step"], d["flow_kernel_ms"], d["pregen_ms"], d["summary_ms"], f["list_entries"], f["ring_rows"], f["lds_bytes_per_wave"], str(f["handed_back"]), d["roofline"]["frac"]))
else: print(tag, "FAILED"); print(open(path).read()[-1500:])
PY
}
i=0
for spec in "$@"; do
  i=$((i+1))
  lib="${spec%%|*}"; flags="${spec#*|}"
  if [ -n "$lib" ]; then export ASYNCFLOW_HIP_LIB=$PWD/asyncflow_amd/csrc/libasyncflow_hip_$lib.so; else unset ASYNCFLOW_HIP_LIB; fi
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --generic-kernels $flags > gpurun_out/ab_$i.log 2>&1
  show "[$lib] $flags" gpurun_out/ab_$i.log
done
