#!/bin/bash
# Round-6 evidence (GPU box, from the repo root):   bash scripts/profile_round6.sh <tag> <step> [<step> ...]
#   tests     python -m pytest tests -m gpu -x -q                                   -> gputests.log
#   driver    the driver's bench command (with the CPU baseline)                    -> bench_driver_cmd.log
#   cal       scripts/microbench/valu_calibration.hip unprofiled + one --pmc pass   -> valu_calibration.json
#   c2|c3|c4|c5|gensrv   for that bench config: an unprofiled line, kernel trace + stats, every --pmc set in its OWN run
#             (never together with trace domains), binding_<label>.json / traffic_<label>.json (scripts/make_binding_json.py)
#   sec2|sec3|sec4|sec5|secgensrv   AF_FLOW_PROF section profile of af_flow_jit     -> flow_sections_<label>.txt
# Everything lands in gpurun_out/prof_<tag>/ ; copy what is to be judged into profiles/r06/.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
KERN="af_flow|af_des|af_jit|af_pregen|af_arrival|af_summary|af_series"
cfg_of() { case $1 in c2) echo 2;; c3) echo 3;; c4) echo 4;; c5) echo 5;; gensrv) echo 6;; esac; }
for step in "$@"; do
  echo "=== $step ($(date +%T))"
  case $step in
    tests)
      ( time timeout 3000 python -m pytest tests -m gpu -q --durations=25 ) > $OUT/gputests.log 2>&1; echo "rc=$?" >> $OUT/gputests.log; tail -45 $OUT/gputests.log ;;
    driver)
      ( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd.log 2>&1; echo "rc=$?" >> $OUT/bench_driver_cmd.log
      grep '^{' $OUT/bench_driver_cmd.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['flow_kernel_ms'], d['roofline']['frac'], d['parity_spot_check']['ok'], d['parity_spot_check']['scenarios'])" ;;
    cal)
      hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_calibration scripts/microbench/valu_calibration.hip 2> /dev/null
      /tmp/valu_calibration > $OUT/cal_stdout.json
      rm -rf /tmp/pcal; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pcal -o c -- /tmp/valu_calibration > $OUT/cal_under_pmc.log 2>&1
      cp $(find /tmp/pcal -name "*counter_collection.csv" | head -1) $OUT/cal_counters.csv
      python scripts/make_valu_calibration.py $OUT | tail -14 ;;
    c2|c3|c4|c5|gensrv)
      L=$step; C=$(cfg_of $L); B="python bench.py --config $C --no-cpu-baseline --no-diagnostics"
      $B --steps 3 --warmup 1 > $OUT/bench_unprofiled_$L.log 2>&1
      rm -rf /tmp/pt; timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- $B --steps 3 --warmup 1 --no-parity-check > $OUT/bench_under_trace_$L.log 2>&1
      cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_trace_$L.csv
      f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1); head -1 $f > $OUT/kernel_trace_af_$L.csv; grep -E "$KERN" $f >> $OUT/kernel_trace_af_$L.csv
      pass() { i=$1; shift; rm -rf /tmp/pp$i; timeout -k 10 420 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pp$i -o p -- $B --steps 1 --warmup 0 --no-parity-check > $OUT/bench_under_pmc${i}_$L.log 2>&1; f=$(find /tmp/pp$i -name "*counter_collection.csv" | head -1); head -1 $f > $OUT/pmc${i}_$L.csv; grep -E "$KERN" $f >> $OUT/pmc${i}_$L.csv; }
      pass 1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES
      pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU
      pass 3 FETCH_SIZE
      pass 4 WRITE_SIZE
      pass 5 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
      python scripts/make_binding_json.py $OUT $L | tail -40 ;;
    sec2|sec3|sec4|sec5|secgensrv)
      L=${step#sec}; [ $L = gensrv ] || L=c$L; C=$(cfg_of $L)
      rm -f $OUT/flow_sections_$L.txt
      AF_FLOW_PROF=$OUT/flow_sections_$L.txt python bench.py --config $C --steps 1 --warmup 0 --no-cpu-baseline --no-diagnostics --no-parity-check > $OUT/prof_$L.log 2>&1
      tail -14 $OUT/flow_sections_$L.txt ;;
  esac
done
ls $OUT | head -80
