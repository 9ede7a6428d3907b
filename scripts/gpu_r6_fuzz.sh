#!/bin/bash
# Round-6 fuzz campaigns on the round's tree (fresh payload ranges): tallies -> gpurun_out/fuzz_r06/*.json
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/fuzz_r06; mkdir -p $OUT
run() { name=$1; shift; echo "=== $name ($(date +%T))"; ( time timeout 1500 python "$@" ) > $OUT/$name.json 2> $OUT/$name.err; tail -c 600 $OUT/$name.json; tail -3 $OUT/$name.err; }
K=${1:-0}
run gpu_fuzz_frac_ram_400_payloads_k$((1000+K)) scripts/gpu_fuzz_frac_ram.py 400 $((1000+K))
run gpu_fuzz_f3_120_payloads_k$((80000+K)) scripts/gpu_fuzz_f3.py 120 $((80000+K))
run gpu_fuzz_sweeps_600_payloads_k$((90000+K)) scripts/gpu_fuzz_sweeps.py 600 $((90000+K))
run gpu_fuzz_analyzer_1500_payloads_k$((100000+K)) scripts/gpu_fuzz_analyzer.py 1500 $((100000+K))
