#!/bin/bash
# Round-6 closing fuzz campaign, twice the payloads of scripts/gpu_r6_fuzz.sh, on payload ranges nothing else used: tallies -> gpurun_out/fuzz_r06/*.json
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/fuzz_r06; mkdir -p $OUT
run() { name=$1; shift; echo "=== $name ($(date +%T))"; ( time timeout 1700 python "$@" ) > $OUT/$name.json 2> $OUT/$name.err; tail -c 300 $OUT/$name.json; tail -3 $OUT/$name.err; }
K=${1:-600000}
run gpu_fuzz_f3_240_payloads_k$((80000+K)) scripts/gpu_fuzz_f3.py 240 $((80000+K))
run gpu_fuzz_sweeps_1200_payloads_k$((90000+K)) scripts/gpu_fuzz_sweeps.py 1200 $((90000+K))
run gpu_fuzz_analyzer_3000_payloads_k$((100000+K)) scripts/gpu_fuzz_analyzer.py 3000 $((100000+K))
run gpu_fuzz_frac_ram_800_payloads_k$((1000+K)) scripts/gpu_fuzz_frac_ram.py 800 $((1000+K))
