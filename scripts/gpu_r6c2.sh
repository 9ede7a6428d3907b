#!/bin/bash
# Round 6, last session: sweeps in flight with staggered starts; the new GPU test.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r6c; mkdir -p $OUT
for spec in "2 2 -1" "2 2 0" "2 3 -1" "3 2 -1" "6 2 -1" "4 2 -1"; do set -- $spec
  timeout 400 python scripts/gpu_two_sweeps_in_flight.py --config $1 --in-flight $2 --stagger $3 --steps 10 > $OUT/sweeps_in_flight_c$1_k$2_stagger$3.json 2> $OUT/sweeps_in_flight_c$1_k$2_stagger$3.err; echo "c$1 k$2 stagger $3 rc=$? $(cat $OUT/sweeps_in_flight_c$1_k$2_stagger$3.json)"
done
timeout 600 python -m pytest tests/test_gpu_flow.py -m gpu -q -k sweeps_in_flight > $OUT/gputest_sweeps_in_flight.log 2>&1; tail -3 $OUT/gputest_sweeps_in_flight.log
