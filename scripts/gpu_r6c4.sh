#!/bin/bash
# Round 6, last session: kernel trace of two sweeps in flight (which kernel of one sweep runs beside which of the other).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r6c_handover; mkdir -p $OUT
KERN="af_flow|af_des|af_jit|af_pregen|af_arrival|af_summary|af_series"
rm -rf /tmp/pt2; timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt2 -o t -- python scripts/gpu_two_sweeps_in_flight.py --config 2 --in-flight 2 --steps 4 > $OUT/sweeps_in_flight_under_trace_c2.log 2>&1
cp $(find /tmp/pt2 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_trace_two_in_flight_c2.csv
f=$(find /tmp/pt2 -name "*kernel_trace.csv" | head -1); head -1 $f > $OUT/kernel_trace_af_two_in_flight_c2.csv; grep -E "$KERN" $f >> $OUT/kernel_trace_af_two_in_flight_c2.csv
tail -2 $OUT/sweeps_in_flight_under_trace_c2.log; wc -l $OUT/kernel_trace_af_two_in_flight_c2.csv
