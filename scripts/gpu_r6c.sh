#!/bin/bash
# Round 6, last session: sweeps in flight (scripts/gpu_two_sweeps_in_flight.py), then the GPU suite and the driver's command on the final tree.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r6c; mkdir -p $OUT
for spec in "2 2" "2 3" "3 2" "6 2" "5 2"; do set -- $spec
  timeout 300 python scripts/gpu_two_sweeps_in_flight.py --config $1 --in-flight $2 --steps 6 > $OUT/sweeps_in_flight_c$1_k$2.json 2> $OUT/sweeps_in_flight_c$1_k$2.err; echo "c$1 k$2 rc=$? $(cat $OUT/sweeps_in_flight_c$1_k$2.json)"
done
bash scripts/profile_round6.sh r6c driver tests
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
