#!/bin/bash
# Round 6, second session: final evidence on the round's last kernel sources -- GPU suite, the driver's command, every config's
# trace + PMC passes + section profile, the step as two calls, without a compiler, as one RCCL rank, on the generic kernels, and
# the fuzz campaigns on fresh payload ranges.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
bash scripts/profile_round6.sh final2 tests driver c2 sec2 c3 sec3 c4 c5 sec5 gensrv secgensrv
OUT=gpurun_out/prof_final2
python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics --separate-summary > $OUT/bench_c2_two_calls.log 2>&1
ASYNCFLOW_NO_HIPCC=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_without_hipcc_prebuilt_kernels.log 2>&1
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 AF_BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_rccl_world1.log 2>&1
python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics --generic-kernels > $OUT/bench_c2_generic_kernels.log 2>&1
python bench.py --config 6 --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics --generic-kernels > $OUT/bench_c6_generic_kernels.log 2>&1
for f in bench_c2_two_calls bench_without_hipcc_prebuilt_kernels bench_rccl_world1 bench_c2_generic_kernels bench_c6_generic_kernels; do
  grep '^{' $OUT/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'value %.4g' % d['value'], 'rccl', d.get('rccl_ranks'), 'jit', d.get('plan_specialised_kernel'), d.get('jit_fallbacks'))"
done
for c in 3 4 5 6; do
  python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics --separate-summary > $OUT/bench_c${c}_two_calls.log 2>&1
  grep '^{' $OUT/bench_c${c}_two_calls.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c two calls', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'value %.4g' % d['value'])"
done
bash scripts/gpu_r6_fuzz.sh ${1:-20000}
