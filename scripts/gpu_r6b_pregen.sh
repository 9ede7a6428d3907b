#!/bin/bash
# af_arrival_groups with 7 / 11 (the tree) / 15 producer waves per workgroup (library variants built in the build container into
# build/variants/) x scenarios per workgroup, BASELINE config 2.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pregen_r06b; mkdir -p $OUT
LIB=asyncflow_amd/csrc/libasyncflow_hip.so
cp $LIB /tmp/lib_tree.so
line() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'pregen %.2f' % d['pregen_ms'], 'group', d.get('pregen_group'), 'parity', d['parity_spot_check']['ok'])"; }
for rep in 1 2; do
  for v in tree p15 p7; do
    if [ $v = tree ]; then cp /tmp/lib_tree.so $LIB; else cp build/variants/libaf_$v.so $LIB; fi
    for g in 40 48 64; do
      [ $v = p7 ] && [ $g != 40 ] && continue
      AF_PREGEN_GROUP=$g timeout 600 python bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline --no-diagnostics --separate-summary > $OUT/c2_${v}_g${g}_$rep.log 2>&1; line $OUT/c2_${v}_g${g}_$rep.log "producers $v group $g"
    done
  done
done
cp /tmp/lib_tree.so $LIB
