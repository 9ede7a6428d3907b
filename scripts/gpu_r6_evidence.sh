#!/bin/bash
# Round-6 evidence call: the two tests fixed after the first full-suite run, then the profile passes of every bench config.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof_b
( time timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_flow.py -m gpu -q -k "waiting_ram_puts or general_server_benchmark_batch" ) > gpurun_out/prof_b/gputests_fixed.log 2>&1
tail -5 gpurun_out/prof_b/gputests_fixed.log
bash scripts/profile_round6.sh b c2 sec2 c3 c4 c5 gensrv
