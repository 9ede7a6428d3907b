#!/bin/bash
# Round 6 (second session): the general server station's all-pairs loops with their LDS reads fetched ahead -- config 6 bench,
# section profile, and the GPU tests that hold the station to the next-event kernels / the oracle.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/gensrv_r06b; mkdir -p $OUT
line() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', 'ms/step %.2f' % d['ms_per_step'], 'flow %.2f' % d['flow_kernel_ms'], 'value %.4g' % d['value'], 'parity', d.get('parity_spot_check',{}).get('ok'))"; }
timeout 900 python bench.py --config 6 --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > $OUT/bench_c6.log 2>&1; line $OUT/bench_c6.log "c6"
rm -f $OUT/flow_sections_c6.txt
AF_FLOW_PROF=$OUT/flow_sections_c6.txt timeout 600 python bench.py --config 6 --steps 1 --warmup 0 --no-cpu-baseline --no-diagnostics --no-parity-check --separate-summary > $OUT/prof_c6.log 2>&1
tail -13 $OUT/flow_sections_c6.txt
( time timeout 1500 python -m pytest tests/test_gpu_flow.py tests/test_gpu_full_batches.py tests/test_gpu_fuzz.py -m gpu -q -x -k "general or gensrv or endpoint or tiers or f3 or feed" ) > $OUT/gputests_gensrv.log 2>&1; tail -5 $OUT/gputests_gensrv.log
( time timeout 900 python scripts/gpu_fuzz_f3.py 40 120000 ) > $OUT/gpu_fuzz_f3_40_payloads_k120000.json 2> $OUT/gpu_fuzz_f3.err; tail -c 600 $OUT/gpu_fuzz_f3_40_payloads_k120000.json; tail -4 $OUT/gpu_fuzz_f3.err
