#!/bin/bash
# Round-3 rocprofv3 evidence (GPU box, from the repo root): bash scripts/profile_round3.sh [tag] [bench flags...]
#   -> gpurun_out/prof_<tag>/*   kernel trace + stats of the default bench command; every --pmc set in its OWN run
#      (never together with trace domains); binding.json / traffic.json by scripts/make_binding_json.py
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r03}; shift
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
KERN="af_flow|af_des|af_jit|af_pregen|af_arrival|af_summary|af_series"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics "$@" > $OUT/bench_unprofiled.log 2>&1      # (also fills the JIT cache)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-diagnostics "$@" > $OUT/bench_under_trace.log 2>&1
cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_trace.csv
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1); head -1 $f > $OUT/kernel_trace_af.csv; grep -E "$KERN" $f >> $OUT/kernel_trace_af.csv
EXTRA="$@"
pass() { i=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pp$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check --no-diagnostics $EXTRA > $OUT/bench_under_pmc$i.log 2>&1; f=$(find /tmp/pp$i -name "*counter_collection.csv" | head -1); head -1 $f > $OUT/pmc$i.csv; grep -E "$KERN" $f >> $OUT/pmc$i.csv; }
pass 1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES
pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU
pass 3 FETCH_SIZE
pass 4 WRITE_SIZE
pass 5 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
python scripts/make_binding_json.py $OUT
ls -la $OUT
