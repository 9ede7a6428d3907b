#!/bin/bash
# A/B: producer waves per workgroup of af_arrival_groups (variant libraries built with -DAF_PREGEN_PRODUCERS=n under
# asyncflow_amd/csrc/_ab/), and the grouped pre-generation forced on the users x RTT grids (configs 3 / 4: consecutive
# scenarios share their users value, so a workgroup's scenarios ARE alike).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/ab2; mkdir -p $O
run() { # tag, env..., -- bench flags
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --no-parity-check "$@" > $O/$tag.log 2>&1
  python - $tag $O/$tag.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("%-14s pregen %.2f flow %.2f ms/step %.2f" % (sys.argv[1], d["pregen_ms"], d["flow_kernel_ms"], d["ms_per_step"]))
else: print(sys.argv[1], "FAILED", open(sys.argv[2]).read()[-600:])
PY
}
run base X=1 --
for P in 13 15; do
  L=asyncflow_amd/csrc/_ab/lib_p$P.so
  [ -f $L ] || { echo "no $L"; continue; }
  run p$P ASYNCFLOW_HIP_LIB=$PWD/$L --
  for G in "$@"; do run p${P}_g$G ASYNCFLOW_HIP_LIB=$PWD/$L AF_PREGEN_GROUP=$G --; done
done
run c3_rows X=1 -- --config 3
run c3_groups AF_PREGEN_MODE=groups -- --config 3
run c5_base X=1 -- --config 5
[ -f asyncflow_amd/csrc/_ab/lib_p15.so ] && run c5_p15 ASYNCFLOW_HIP_LIB=$PWD/asyncflow_amd/csrc/_ab/lib_p15.so -- --config 5
run c4_groups AF_PREGEN_MODE=groups -- --config 4
