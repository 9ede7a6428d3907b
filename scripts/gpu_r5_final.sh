#!/bin/bash
# Round 5 evidence on the final sources: the GPU suite, the driver's command, the VALU calibration, trace + PMC passes + binding
# of every bench config, section profiles, the GPU fuzz, server tiers, the bench without a compiler.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05; mkdir -p $OUT
bash scripts/profile_round5.sh r05 tests driver cal c2 c3 c4 c5 gensrv sec2 sec5 secgensrv
python scripts/gpu_fuzz_f3.py 24 > $OUT/gpu_fuzz_f3.json 2> $OUT/gpu_fuzz_f3.err; cat $OUT/gpu_fuzz_f3.json | cut -c1-1200
python scripts/gpu_chain.py 10000 600 > $OUT/chain_shared_backend_10000_T600.json 2> $OUT/chain.err; cut -c1-400 $OUT/chain_shared_backend_10000_T600.json
ASYNCFLOW_NO_HIPCC=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_without_hipcc_prebuilt_kernels.log 2>&1
grep '^{' $OUT/bench_without_hipcc_prebuilt_kernels.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no hipcc:', d['ms_per_step'], d['config']['flow']['plan_specialised_kernel'], d['config']['flow']['jit_fallbacks'])"
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 AF_BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics > $OUT/bench_rccl_world1.log 2>&1
grep '^{' $OUT/bench_rccl_world1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rccl world 1:', d['rccl_ranks'], d['per_rank'], d['gather_ms'])"
ls $OUT | wc -l
