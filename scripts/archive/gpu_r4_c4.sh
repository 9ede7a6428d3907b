#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_c4; mkdir -p $OUT
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_config4.log 2>&1
grep '^{' $OUT/bench_config4.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['config']['flow']['handed_back']['total'], d['parity_spot_check']['ok'], d['config']['flow']['jit_fallbacks'])"
timeout 900 python -m pytest tests/test_gpu_flow.py -m gpu -x -q -k "events or spike or grid_corners or sweep_over or negative" 2>&1 | tail -2
